"""Phase timeline of workgroup 0 of escore_mol_fwd_kernel (library built with MSDE_HIPCC_FLAGS=-DES_TIMING=1)."""
import ctypes, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import moleculesde_amd.geom3d as G
from moleculesde_amd import plan as P, _lib
from moleculesde_amd.geom3d import sde_2d_to_3d as M
from moleculesde_amd.batch import Batch
from moleculesde_amd.synthetic import make_batch, make_molecule
dev = torch.device("cuda", 0)
torch.manual_seed(0)
lib = _lib.load()
fn = lib.msde_escore_debug_stamps
fn.argtypes = [ctypes.c_void_p]
names = {0: "start", 40: "end"}
for l in range(4):
    for k, nm in ((1, "staged+W->LDS"), (2, "qkvs"), (3, "edge proj"), (4, "attention"), (5, "tail"), (6, "basis mlp")):
        names[k + 8 * l] = f"L{l} {nm}"
for name, cpu_b in (("10x14", Batch.from_data_list([make_molecule(np.random.default_rng(0), 14)] * 10)), ("batch256", make_batch(256, 0))):
    b = G.prepare_batch(cpu_b.clone(), dev)
    pl = P.get_plan(b); ep = pl.ext
    net = M.EquivariantScoreNetwork(32, hidden_coff_dim=128).to(dev).eval()
    x, ea, bs = torch.randn(ep.N, 32, device=dev), torch.randn(ep.E, 32, device=dev), torch.randn(ep.E, 9, device=dev)
    with torch.no_grad():
        for _ in range(5):
            net(ep, x, ea, bs, pl)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    assert fn(buf) == 0
    st = {i: buf[i] for i in names}
    t0 = st[0]
    mp = pl.mol_ptr.cpu(); rp = ep.rowptr.cpu()
    print(f"--- {name}: molecule 0 has {int(mp[1])} atoms, {int(rp[int(mp[1])])} edges")
    prev = t0
    for i in sorted(names):
        if st[i] >= t0:
            print(f"{names[i]:24s} +{(st[i] - prev) / 100:7.2f} us   t={(st[i] - t0) / 100:7.2f}")
            prev = st[i]
