"""Phase timeline of workgroup 0 of the escore_mol kernels (library built with MSDE_HIPCC_FLAGS=-DES_TIMING=1).
usage: escore_phases.py [atoms of the single molecule]"""
import ctypes, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import moleculesde_amd.geom3d as G
from moleculesde_amd import plan as P, _lib
from moleculesde_amd.geom3d import sde_2d_to_3d as M
from moleculesde_amd.batch import Batch
from moleculesde_amd.synthetic import make_molecule
dev = torch.device("cuda", 0)
M_ = __import__("moleculesde_amd.geom3d.sde_2d_to_3d", fromlist=["x"]); M_.MOL_KERNEL_TRAIN = True
torch.manual_seed(0)
lib = _lib.load()
fn = lib.msde_escore_debug_stamps
fn.argtypes = [ctypes.c_void_p]
names = {0: "fwd start", 40: "fwd end", 64: "bwd setup"}
for l in range(4):
    for k, nm in ((1, "staged+W->LDS"), (2, "qkvs"), (3, "edge proj"), (4, "attention"), (5, "tail"), (6, "basis mlp")):
        names[k + 8 * l] = f"fwd L{l} {nm}"
    b = 14 * (3 - l)
    for k, nm in ((65, "basis prep"), (66, "basis P"), (67, "basis sweep A"), (68, "basis gH/gW1a"), (69, "basis sweep B"),
                  (70, "tail T4'"), (71, "tail T3'+T2'"), (72, "tail T1'+qkvs"), (73, "edge proj"), (74, "softmax bwd"),
                  (75, "gee products + k/v"), (76, "qkvs bwd")):
        names[k + b] = f"bwd L{l} {nm}"
natoms = int(sys.argv[1]) if len(sys.argv) > 1 else 14
cpu_b = Batch.from_data_list([make_molecule(np.random.default_rng(0), natoms)] * 4)
b = G.prepare_batch(cpu_b.clone(), dev)
pl = P.get_plan(b); ep = pl.ext
net = M.EquivariantScoreNetwork(32, hidden_coff_dim=128).to(dev).train()
x = torch.randn(ep.N, 32, device=dev, requires_grad=True); ea = torch.randn(ep.E, 32, device=dev, requires_grad=True)
bs = torch.randn(ep.E, 9, device=dev)
for _ in range(4):
    net.zero_grad(set_to_none=True)
    net(ep, x, ea, bs, pl)["gradient"].sum().backward()
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 128)()
assert fn(buf) == 0
buf2 = (ctypes.c_longlong * 128)()
fn2 = lib.msde_escore_debug_stamps_bwd
fn2.argtypes = [ctypes.c_void_p]
assert fn2(buf2) == 0
for i in range(64, 128):
    buf[i] = buf2[i]
mp = pl.mol_ptr.cpu(); rp = ep.rowptr.cpu()
print(f"--- molecule 0 has {int(mp[1])} atoms, {int(rp[int(mp[1])])} edges")
for lo, hi, label in ((0, 63, "forward"), (64, 127, "backward")):
    idx = [i for i in sorted(names) if lo <= i <= hi and buf[i] > 0]
    t0 = prev = buf[idx[0]]
    for i in idx:
        # the backward walks layers 3..0: order by time
        pass
    for i in sorted(idx, key=lambda i: buf[i]):
        print(f"{names[i]:34s} +{(buf[i] - prev) / 100:7.2f} us   t={(buf[i] - t0) / 100:7.2f}")
        prev = buf[i]
