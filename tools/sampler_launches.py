"""Kernel launches of ONE predictor-corrector iteration of the 2D->3D sampler (eager, torch profiler) and the time of the
replayed hipGraph per iteration with / without the per-iteration host copy."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import moleculesde_amd.geom3d as G
from moleculesde_amd import sampler
from moleculesde_amd.batch import Batch
from moleculesde_amd.synthetic import make_molecule
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
torch.manual_seed(0)
mol = make_molecule(np.random.default_rng(0), 14)
b = G.prepare_batch(Batch.from_data_list([mol] * 10), dev)
gnn = G.GNN(5, 300, JK="last", drop_ratio=0, gnn_type="GIN").to(dev).eval()
s23 = G.SDEModel2Dto3D_02(emb_dim=300, hidden_dim=32, beta_min=0.2, beta_max=1.0, num_diffusion_timesteps=1000,
                          beta_schedule=None, SDE_type="VE", use_extend_graph=True).to(dev).eval()
with torch.no_grad():
    rep = gnn(b.x, b.edge_index, b.edge_attr)
sampler.position_PC_generation(s23, rep, b, num_steps=4, use_graph=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    sampler.position_PC_generation(s23, rep, b, num_steps=3, use_graph=False)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA]
tot = 0
for e in sorted(rows, key=lambda e: -e.count):
    print(f"{e.count / 3:6.1f} x {e.device_time_total / max(e.count, 1):8.1f} us  {e.key[:110]}")
    tot += e.count
print("kernels per PC iteration:", tot / 3)
