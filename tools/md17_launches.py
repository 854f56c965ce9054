"""Kernel census of ONE MD17 force fine-tuning step (21 atoms, batch 1; eager, torch profiler)."""
import os, sys, torch
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
R = 3
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(R):
        ft.step(b, et, ftg)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA]
tot = 0
tt = 0.0
for e in sorted(rows, key=lambda e: -e.count):
    print(f"{e.count / R:6.1f} x {e.device_time_total / max(e.count, 1):8.1f} us  {e.key[:150]}")
    tot += e.count
    tt += e.device_time_total
print("kernels per step:", tot / R, " sum of kernel time per step:", tt / R, "us")
