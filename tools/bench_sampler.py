"""BASELINE.json config 4: 2D->3D VE reverse-SDE sampling, one molecule x 10 replicas, predictor +
corrector per step; reports steps/s eager vs hipGraph replay (GPU box)."""
import os, sys, time, torch, numpy as np
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
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for use_graph in (False, True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pos = sampler.position_PC_generation(s23, rep, b, num_steps=steps, use_graph=use_graph)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"use_graph={use_graph}: {steps} PC steps ({2 * steps} score calls) in {dt:.2f}s = {steps / dt:.0f} steps/s; "
          f"finite={bool(torch.isfinite(pos).all())}; atoms={pos.size(0)}")
