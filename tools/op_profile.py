"""torch.profiler view of ONE eager pretrain step: which aten ops (glue around the HIP kernels) still launch
kernels, with shapes and the Python call site.  GPU box only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from moleculesde_amd import pretrain
from moleculesde_amd.geom3d import prepare_batch
from moleculesde_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
torch.manual_seed(0)
args = pretrain.readme_args()
if "--full" not in sys.argv:
    args.SDE_coeff_generative_3Dto2D = 0
tr = pretrain.Trainer(args, dev)
b = prepare_batch(make_batch(256, seed=0), dev)
for _ in range(3):
    tr.step(b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(b)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=6)
rows = []
for e in ka:
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0 and e.key.startswith("aten::"):
        stack = [s for s in e.stack if "moleculesde_amd" in s or "bench.py" in s]
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:70], (stack[0] if stack else "")[-80:]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"aten ops with device time: {len(rows)} groups, total {tot:.0f} us")
for dt, c, k, sh, st in rows[:70]:
    print(f"{dt:8.1f} us x{c:3d} {k:28s} {sh:70s} {st}")
