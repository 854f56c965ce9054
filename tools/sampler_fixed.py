"""Where the sampler's fixed cost goes: cProfile of a 3-iteration call (tabulation, two eager iterations, capture, one replay)."""
import os, sys, time, torch, numpy as np, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import moleculesde_amd.geom3d as G
from moleculesde_amd import sampler
from moleculesde_amd.batch import Batch
from moleculesde_amd.synthetic import make_molecule
dev = torch.device("cuda", 0)
torch.manual_seed(0)
rng = np.random.default_rng(0)
mol = make_molecule(rng, 14)
b = G.prepare_batch(Batch.from_data_list([mol] * 10), dev)
gnn = G.GNN(5, 300, JK="last", drop_ratio=0, gnn_type="GIN").to(dev).eval()
s23 = G.SDEModel2Dto3D_02(emb_dim=300, hidden_dim=32, beta_min=0.2, beta_max=1.0, num_diffusion_timesteps=1000,
                          beta_schedule=None, SDE_type="VE", use_extend_graph=True).to(dev).eval()
with torch.no_grad():
    rep = gnn(b.x, b.edge_index, b.edge_attr)
sampler.position_PC_generation(s23, rep, b, num_steps=50)
for steps in (3, 3, 1000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sampler.position_PC_generation(s23, rep, b, num_steps=steps)
    torch.cuda.synchronize(); print(steps, "iterations:", (time.perf_counter() - t0) * 1e3, "ms")
pr = cProfile.Profile()
pr.enable()
sampler.position_PC_generation(s23, rep, b, num_steps=3)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
