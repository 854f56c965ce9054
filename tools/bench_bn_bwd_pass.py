"""Launch time of the BatchNorm-backward column pass of a GIN layer: finish + pass as two launches vs msde_bn_bwd_fin_cols."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip, _lib
dev = torch.device("cuda", 0)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, C in [(3588, 600), (3588, 300)]:
    strips = (M + 63) // 64
    stats = torch.randn(strips, 2, C, device=dev)
    gamma, mean, rstd = torch.randn(C, device=dev), torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
    G, Z, out = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev), torch.empty(M, C, device=dev)

    def two():
        pw, gb = hip._bn_fin_bwd(stats, strips, M, C, gamma, mean, rstd)
        _lib.call("msde_bn_bwd_cols", hip._p(G), C, hip._p(Z), C, hip._p(pw[0]), hip._p(pw[1]), hip._p(pw[2]), hip._p(None), hip._p(None),
                  M, hip._p(None), C, hip._p(out), C, hip._stream())

    def one():
        hip._bn_bwd_fin_cols(stats, strips, M, C, gamma, mean, rstd, G, Z, None, None, out, None)
    print(f"{M} x {C}: finish + pass {timed(two):.1f} us, fused {timed(one):.1f} us (back to back in a graph)")
