"""Shapes of the additions autograd's gradient accumulation launches in one MD17 step (which tensors have several consumers)."""
import os, sys, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import moleculesde_amd.geom3d as G
from moleculesde_amd.synthetic import make_md17_batch
from moleculesde_amd.finetune_md17 import ForceTrainer
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
torch.manual_seed(0)
kw = dict(hidden_channels=300, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=10, readout="mean", node_class=119)
cpu_b = make_md17_batch(1, seed=3, n_atoms=21)
sch, head = G.SchNet(**kw).to(dev), torch.nn.Linear(300, 1).to(dev)
b = G.prepare_batch(cpu_b.clone(), dev)
ft = ForceTrainer(sch, head, lr=5e-4, energy_coeff=1.0, force_coeff=1.0)
et, ftg = torch.randn(1, device=dev), torch.randn(21, 3, device=dev)
for _ in range(3):
    ft.step(b, et, ftg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    ft.step(b, et, ftg)
    torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.name in ("aten::add", "aten::add_", "aten::mul", "aten::neg", "aten::clone", "aten::copy_", "aten::sum", "aten::zeros_like", "aten::fill_", "aten::zero_"):
        c[(e.name, str(e.input_shapes))] += 1
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    print(f"{v:4d} x {k[0]:14s} {k[1]}")
