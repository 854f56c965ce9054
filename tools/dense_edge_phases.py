"""Phase timeline of workgroups 0 (node half) and 1 (pair half) of dense_edge_layer_fwd_kernel in a --full training step
(library built with MSDE_HIPCC_FLAGS=-DDH_TIMING=1)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import _lib, pretrain
from moleculesde_amd.geom3d import prepare_batch
from moleculesde_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
tr = pretrain.Trainer(pretrain.readme_args(SDE_coeff_generative_3Dto2D=1), dev)
bt = prepare_batch(make_batch(256, seed=0), dev)
for _ in range(3):
    tr.step(bt)
torch.cuda.synchronize()
fn = _lib.load().msde_dense_debug_stamps
fn.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_longlong * 64)()
assert fn(buf) == 0
names = {0: "start", 1: "weights -> LDS", 2: "operands staged (+ degree norms)", 3: "GCN", 4: "channel MLP + x_out (node half end)",
         5: "attention products + tanh", 6: "T -> LDS", 7: "pair MLP + stores (pair half end)"}
for wg, label in ((0, "node half (workgroup 0)"), (1, "pair half (workgroup 1)")):
    idx = [i for i in range(8) if buf[32 * wg + i] > 0]
    print(label)
    prev = t0 = buf[32 * wg + idx[0]]
    for i in idx:
        v = buf[32 * wg + i]
        print(f"   {names[i]:40s} +{(v - prev) / 100:7.2f} us   t={(v - t0) / 100:7.2f}")
        prev = v
