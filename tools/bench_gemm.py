"""Micro-benchmark: moleculesde_amd's fp32 MFMA Linear vs the vendor GEMM (torch F.linear) on the
shapes of the MoleculeSDE pretrain step (bs 256).  Run on the GPU box:  python tools/bench_gemm.py"""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip  # noqa: E402

SHAPES = [(3588, 600, 300), (3588, 300, 600), (3588, 128, 300), (3588, 300, 128), (3588, 300, 300),
          (49090, 128, 51), (49090, 128, 128), (35186, 32, 300), (35186, 32, 128), (35186, 128, 64), (35186, 32, 32), (3588, 32, 32), (3588, 128, 32), (3588, 32, 300), (35186, 3, 128), (35186, 66, 32)]


def t(fn, it=30):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / it * 1e3


def main():
    dev = torch.device("cuda", 0)
    hip.set_linear_mode("hip")   # the "hip" columns time csrc/linear.hip regardless of the dispatch policy
    print(f"{'M':>6} {'N':>4} {'K':>4} | {'fwd hip':>8} {'fwd lib':>8} | {'dgrad hip':>9} {'lib':>7} | {'wgrad hip':>9} {'lib':>7} | TF(hip fwd)")
    for M, N, K in SHAPES:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev)
        b = torch.randn(N, device=dev)
        gy = torch.randn(M, N, device=dev)
        xg = x.clone().requires_grad_(True)
        wg = w.clone().requires_grad_(True)
        f_h = t(lambda: hip.linear(x, w, b))
        f_l = t(lambda: F.linear(x, w, b))
        yh = hip.linear(xg, w, None)
        d_h = t(lambda: torch.autograd.grad(yh, xg, gy, retain_graph=True))
        yl = F.linear(xg, w, None)
        d_l = t(lambda: torch.autograd.grad(yl, xg, gy, retain_graph=True))
        yh2 = hip.linear(x, wg, None)
        w_h = t(lambda: torch.autograd.grad(yh2, wg, gy, retain_graph=True))
        yl2 = F.linear(x, wg, None)
        w_l = t(lambda: torch.autograd.grad(yl2, wg, gy, retain_graph=True))
        print(f"{M:6d} {N:4d} {K:4d} | {f_h:8.1f} {f_l:8.1f} | {d_h:9.1f} {d_l:7.1f} | {w_h:9.1f} {w_l:7.1f} | {2.0*M*N*K/f_h/1e6:6.1f}")


if __name__ == "__main__":
    main()
