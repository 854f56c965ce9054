"""Phase timeline of workgroup 0 of msde_escore_mol_score on the sampler's shape (10 replicas of one molecule); library built
with MSDE_HIPCC_FLAGS=-DES_TIMING=1.  usage: escore_score_phases.py [atoms]"""
import ctypes, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import moleculesde_amd.geom3d as G
from moleculesde_amd import _lib
from moleculesde_amd.batch import Batch
from moleculesde_amd.synthetic import make_molecule
dev = torch.device("cuda", 0)
torch.manual_seed(0)
lib = _lib.load()
fn = lib.msde_escore_debug_stamps
fn.argtypes = [ctypes.c_void_p]
names = {0: "start", 40: "end", 41: "staging + geometry prologue", 50: "A start", 51: "A weights requested, staged to LDS", 52: "A barrier",
         53: "A edge features", 54: "A 24 output tiles stored"}
for l in range(4):
    for k, nm in ((1, "layer regs + barrier"), (2, "qkvs"), (3, "edge proj"), (4, "attention"), (5, "tail"), (6, "basis mlp")):
        names[k + 8 * l] = f"L{l} {nm}"
natoms = int(sys.argv[1]) if len(sys.argv) > 1 else 14
b = G.prepare_batch(Batch.from_data_list([make_molecule(np.random.default_rng(0), natoms)] * 10), dev)
gnn = G.GNN(5, 300, JK="last", drop_ratio=0, gnn_type="GIN").to(dev).eval()
s23 = G.SDEModel2Dto3D_02(emb_dim=300, hidden_dim=32, beta_min=0.2, beta_max=1.0, num_diffusion_timesteps=1000,
                          beta_schedule=None, SDE_type="VE", use_extend_graph=True).to(dev).eval()
with torch.no_grad():
    rep = gnn(b.x, b.edge_index, b.edge_attr)
    for _ in range(5):
        s23.get_score_raw(rep, b, b.positions)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 128)()
assert fn(buf) == 0
for lo, hi in ((50, 59), (0, 49)):
    idx = [i for i in sorted(names) if buf[i] > 0 and lo <= i <= hi]
    if not idx:
        continue
    t0 = prev = min(buf[i] for i in idx)
    for i in sorted(idx, key=lambda i: buf[i]):
        print(f"{names[i]:38s} +{(buf[i] - prev) / 100:7.2f} us   t={(buf[i] - t0) / 100:7.2f}")
        prev = buf[i]
