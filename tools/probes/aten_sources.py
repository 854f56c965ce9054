"""Where do the remaining torch operator launches of one pretrain step come from?  Eager step under torch.profiler with
Python stacks: every CPU op that launched a non-library kernel, grouped by (op, innermost moleculesde_amd frame)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from moleculesde_amd import pretrain, bucket as BK
from moleculesde_amd.geom3d import prepare_batch
from moleculesde_amd.synthetic import make_batch
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
torch.manual_seed(0)
full = "--full" in sys.argv
tr = pretrain.Trainer(pretrain.readme_args() if full else pretrain.readme_args(SDE_coeff_generative_3Dto2D=0), dev)
b = prepare_batch(make_batch(256, seed=0), dev)
for _ in range(3):
    tr.step(b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(b)
    torch.cuda.synchronize()
ev = prof.events()
# CPU ops with a stack that directly launched kernels
agg = collections.Counter()
order = []
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    if not e.name.startswith("aten::") and "Memcpy" not in e.name and "Memset" not in e.name:
        continue
    if e.name in ("aten::addmm", "aten::mm"):
        continue
    knames = ",".join(sorted({k.name.split("<")[0].split("(")[0][-28:] for k in e.kernels}))
    chain, p = [], e.cpu_parent
    while p is not None:
        chain.append(p.name.replace("autograd::engine::evaluate_function: ", "bwd:"))
        p = p.cpu_parent
    shapes = str(e.input_shapes)[:60]
    key = (e.name, shapes, " < ".join(chain[:3])[:80])
    agg[key] += len(e.kernels)
    order.append((e.time_range.start, key))
seen = set()
tot = 0
for t, key in sorted(order):
    if key in seen:
        continue
    seen.add(key)
    tot += agg[key]
    print(f"{agg[key]:3d} {key[0]:22s} {key[1]:60s} {key[2]}")
print("total operator launches (without mm/addmm):", tot)
