"""GPU-time (hipGraph replay, no host launch overhead) of weight-gradient GEMMs: csrc/linear.hip split-M
kernel (+ fused bias grad) vs vendor mm (+ msde_colsum), and forward / dgrad, on the pretrain-step shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip, _lib
from moleculesde_amd import slabs  # noqa: E402
dev = torch.device("cuda", 0)
SHAPES = [(3588, 300, 300), (3588, 600, 300), (3588, 300, 600), (3588, 128, 300), (3588, 300, 128), (3588, 32, 32),
          (3588, 128, 32), (3588, 32, 128), (3588, 32, 300), (35186, 32, 32), (35186, 128, 64), (35186, 64, 128),
          (35186, 32, 300), (35186, 3, 128), (35186, 66, 32), (49090, 128, 128), (256, 300, 300)]
REP = 20
if "--node" in sys.argv:
    SHAPES = [(3588, 300, 300), (3588, 600, 300), (3588, 300, 600), (3588, 300, 128), (3588, 128, 300)]
def gtime(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP): fn()
        g.replay(); s.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(5): g.replay()
        b.record(s); b.synchronize()
    return a.elapsed_time(b) / (5 * REP) * 1e3
p = hip._p
print(f"{'M':>6} {'N':>4} {'K':>4} | wgrad hip   lib+colsum | fwd hip    lib | dgrad hip   lib")
for M, N, K in SHAPES:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    g = torch.randn(M, N, device=dev)
    gw = torch.empty(N, K, device=dev); gb = torch.empty(N, device=dev); y = torch.empty(M, N, device=dev)
    gx = torch.empty(M, K, device=dev)
    ws = slabs._wgrad_workspace(M, N, K, dev); bws = hip._bn_workspace(M, N, dev)
    def w_hip(): _lib.call("msde_linear_bwd_w", p(g), p(x), M, N, K, p(gw), p(gb), p(ws), p(None), hip._stream())
    def w_lib():
        torch.mm(g.t(), x, out=gw)
        _lib.call("msde_colsum", p(g), M, N, p(gb), p(bws), p(None), hip._stream())
    def f_hip(): _lib.call("msde_linear_fwd", p(x), p(w), p(b), M, N, K, p(y), hip._stream())
    def f_lib(): torch.addmm(b, x, w.t(), out=y)
    def d_hip(): _lib.call("msde_linear_bwd_x", p(g), p(w), M, N, K, p(gx), hip._stream())
    def d_lib(): torch.mm(g, w, out=gx)
    if "--node" in sys.argv or "--wgrad" in sys.argv:
        print(f"{M:6d} {N:4d} {K:4d} | wgrad hip {gtime(w_hip):8.1f}", flush=True)
        continue
    print(f"{M:6d} {N:4d} {K:4d} | {gtime(w_hip):8.1f} {gtime(w_lib):10.1f} | {gtime(f_hip):6.1f} {gtime(f_lib):6.1f} | {gtime(d_hip):6.1f} {gtime(d_lib):6.1f}", flush=True)
