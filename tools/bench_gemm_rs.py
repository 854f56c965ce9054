"""Time msde_gemm_rs (csrc/gemm_rs.hip) against msde_gemm_ex and the vendor GEMM (torch.addmm / torch.mm) on the dense
shapes of the step, both weight layouts.  hipGraph-timed back-to-back launches; TFLOP/s against the 157.3 fp32 peak.
Environment: MSDE_RS_RT / MSDE_RS_SPLITS / MSDE_RS_KERNEL override the geometry (tuning)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from moleculesde_amd import hip
from bench_gemm_ex import timeit

dev = torch.device("cuda", 0)
SHAPES = [(3588, 300, 300), (3588, 600, 300), (3588, 300, 600), (3588, 128, 300), (3588, 300, 128), (3588, 32, 300),
          (3588, 128, 32), (35186, 32, 300), (35186, 128, 32), (35186, 32, 128), (35186, 32, 64), (35186, 300, 32),
          (35186, 128, 64), (3588, 728, 364), (3588, 728, 728), (3588, 119, 728), (49090, 128, 128)]
if __name__ == "__main__":
    only = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
    for M, N, K in (only or SHAPES):
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        Wk = W.t().contiguous()
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        fl = 2.0 * M * N * K
        row = f"M={M:6d} N={N:4d} K={K:4d}"
        for km in (True,):
            Bop = Wk if km else W
            try:
                t_rs = timeit(lambda: hip.gemm_rs(A, Bop, out, bias=b, b_kmajor=km, fallback=False))
            except Exception as e:
                t_rs = float("nan")
            t_ex = timeit(lambda: hip.gemm_ex(A, Bop, out, bias=b, b_kmajor=km))
            t_lib = timeit((lambda: torch.addmm(b, A, Wk, out=out)) if km else (lambda: torch.addmm(b, A, W.t(), out=out)))
            row += f" | {'KN' if km else 'NK'}: rs {t_rs:6.1f} us ({fl / t_rs / 1e6:5.1f} TF) ex {t_ex:6.1f} lib {t_lib:6.1f}"
        print(row, flush=True)
