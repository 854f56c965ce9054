"""Where does gemm_ex spend its time?  Same launch with (a) everything, (b) no global loads / LDS stores after the first
tile (pure LDS-read + MFMA loop + barriers), (c) additionally no barriers.  Graph-timed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip
from tools.bench_gemm_ex import timeit
dev = torch.device("cuda", 0)
for M, N, K in [(3588, 728, 728), (3588, 728, 364), (3588, 300, 600), (49090, 128, 128)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    r = []
    for flags in (0, 256, 256 | 512, 256 | 512 | 1024):
        t = timeit(lambda: hip.gemm_ex(A, W, out, bias=b, act="silu", _debug_flags=flags))
        r.append(f"{t:6.1f} us ({fl / t / 1e6:5.1f} TF)")
    t = timeit(lambda: torch.addmm(b, A, W.t(), out=out))
    print(f"M={M} N={N} K={K}: full {r[0]} | no loads {r[1]} | no loads, no barriers {r[2]} | + no epilogue stores {r[3]} | library {t:6.1f} us", flush=True)
