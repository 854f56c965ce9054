"""Check and time msde_gemm_t2 (csrc/gemm_t2.hip) against the row-strip kernel and the vendor GEMM on the node-level shapes
of the step, over its tuning space (rows per tile, ring depth, loader wave, column splits).  hipGraph-timed back-to-back
launches; TFLOP/s against the 157.3 fp32 peak.  Usage: bench_gemm_t2.py [MxNxK ...] [--sweep]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from moleculesde_amd import hip
from bench_gemm_ex import timeit

dev = torch.device("cuda", 0)
SHAPES = [(3588, 300, 600), (3588, 600, 300), (3588, 300, 300), (3588, 128, 300), (3588, 300, 128), (3588, 728, 728),
          (3588, 728, 364), (3588, 32, 300), (3588, 600, 600)]


def tune(nw=0, nbuf=0, loader=None):
    return nw + 16 * nbuf + (256 * (loader + 1) if loader is not None else 0)


def check(M, N, K, rt=0, splits=0):
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    ref = (torch.relu(A.double() @ W.double().t() + b.double()) + res.double())
    err = 0.0
    for _ in range(4):          # a race would not show on every launch
        out = torch.full((M, N), float("nan"), device=dev)
        hip.gemm_rs(A, W, out, bias=b, res=res, act="relu", t2=True, rt=rt, splits=splits)
        e = (out.double() - ref).abs().max().item() / ref.abs().max().item()
        err = max(err, e if e == e else float("inf"))
    return err


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    sweep = "--sweep" in sys.argv
    shapes = [tuple(int(v) for v in a.split("x")) for a in args] or SHAPES
    for M, N, K in shapes:
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        Wk = W.t().contiguous()
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        fl = 2.0 * M * N * K
        try:
            t_rs = timeit(lambda: hip.gemm_rs(A, Wk, out, bias=b, b_kmajor=True, fallback=False))
        except Exception:
            t_rs = float("nan")
        t_lib = timeit(lambda: torch.addmm(b, A, Wk, out=out))
        err = check(M, N, K)
        t_t2 = timeit(lambda: hip.gemm_rs(A, W, out, bias=b, t2=True))
        print(f"M={M:5d} N={N:4d} K={K:4d} | t2 {t_t2:6.1f} us ({fl / t_t2 / 1e6:5.1f} TF = {fl / t_t2 / 1e6 / 157.3:.2f}) err {err:.1e}"
              f" | rs {t_rs:6.1f} | lib {t_lib:6.1f}", flush=True)
        if sweep:
            ntiles = (N + 15) // 16
            row = "    splits:"
            for S in (1, 2, 3, 4, 5, 6, 8, 10):
                rn = -(-ntiles // S)
                if rn > 12 or S > ntiles:
                    continue
                try:
                    e = check(M, N, K, 0, S)
                    t = timeit(lambda: hip.gemm_rs(A, W, out, bias=b, t2=True, splits=S))
                    row += f"  S={S}: {t:5.1f}" + ("" if e < 1e-5 else f"(ERR {e:.0e})")
                except Exception as ex:
                    row += f"  S={S}: fail"
            print(row, flush=True)
