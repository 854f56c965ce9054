"""Graph-timed BatchNorm forward / backward (csrc/norm.hip) on the GIN shapes, with the bytes each pass moves."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip, _lib
from tools.bench_gemm_ex import timeit
dev = torch.device("cuda", 0)
p = hip._p
for M, C in ((3588, 600), (3588, 300), (3712, 600), (35186, 300)):
    x = torch.randn(M, C, device=dev); g = torch.randn(M, C, device=dev)
    gamma = torch.rand(C, device=dev) + 0.5; beta = torch.randn(C, device=dev)
    rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
    y = torch.empty_like(x); mean = torch.empty(C, device=dev); rstd = torch.empty(C, device=dev)
    dx = torch.empty_like(x); dg = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
    ws = hip._bn_workspace(M, C, dev)
    f = lambda: _lib.call("msde_bn_fwd", p(x), M, C, p(gamma), p(beta), 1e-5, 0.1, p(rm), p(rv), 1, p(y), p(mean), p(rstd), p(ws), p(None), hip._stream())
    b = lambda: _lib.call("msde_bn_bwd", p(g), p(x), p(mean), p(rstd), p(gamma), p(beta), 1, M, C, p(dx), p(dg), p(db), p(ws), p(None), hip._stream())
    tf, tb = timeit(f), timeit(b)
    mb = M * C * 4 / 1e6
    print(f"M={M:6d} C={C:4d}  fwd (stats + apply) {tf:6.1f} us = {3 * mb / tf:6.2f} TB/s of 3x{mb:.1f} MB   "
          f"bwd (partial + apply) {tb:6.1f} us = {5 * mb / tb:6.2f} TB/s of 5x{mb:.1f} MB", flush=True)
