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
    chain_only = sys.argv[1:] == ["chain"]
    only = [] if chain_only else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
    for M, N, K in ([] if chain_only else (only or SHAPES)):
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
    if not only:
        # SchNet node-level chain (msde_gemm_chain) against its three separate row-strip launches
        M, F, Hd = 3588, 128, 300
        agg, h = torch.randn(M, F, device=dev), torch.randn(M, Hd, device=dev)
        W2, Wl, Wn = (torch.randn(n, k, device=dev) / k ** 0.5 for n, k in ((Hd, F), (Hd, Hd), (F, Hd)))
        b2, bl = torch.randn(Hd, device=dev), torch.randn(Hd, device=dev)
        W2t, Wlt, Wnt = W2.t().contiguous(), Wl.t().contiguous(), Wn.t().contiguous()
        a, hn, x1 = torch.empty(M, Hd, device=dev), torch.empty(M, Hd, device=dev), torch.empty(M, F, device=dev)

        def chain():
            hip.gemm_chain(agg, [dict(W=W2t, N=Hd, K=F, bias=b2, act="ssp", out=a),
                                 dict(W=Wlt, N=Hd, K=Hd, bias=bl, res=h, out=hn), dict(W=Wnt, N=F, K=Hd, out=x1)])

        def separate():
            hip.gemm_rs(agg, W2t, a, bias=b2, act="ssp", b_kmajor=True, fallback=False)
            hip.gemm_rs(a, Wlt, hn, bias=bl, res=h, b_kmajor=True, fallback=False)
            hip.gemm_rs(hn, Wnt, x1, b_kmajor=True, fallback=False)

        fl = 2.0 * M * (F * Hd + Hd * Hd + Hd * F)
        tc, ts = timeit(chain), timeit(separate)
        print(f"SchNet node chain 128->300->300->128, M={M}: chained {tc:6.1f} us ({fl / tc / 1e6:5.1f} TF) | three launches {ts:6.1f} us", flush=True)
        gx, ga, gh = torch.empty(M, Hd, device=dev), torch.empty(M, F, device=dev), torch.empty(M, Hd, device=dev)
        g1, g2 = torch.randn(M, F, device=dev), torch.randn(M, Hd, device=dev)

        def chain_b():
            hip.gemm_chain(g1, [dict(W=Wn, N=Hd, K=F, res=g2, out=gh), dict(W=Wl, N=Hd, K=Hd, act="sspo", dact_from=a, out=gx),
                                dict(W=W2, N=F, K=Hd, out=ga)])

        def separate_b():
            hip.gemm_rs(g1, Wn, gh, res=g2, b_kmajor=True, fallback=False)
            hip.gemm_rs(gh, Wl, gx, act="sspo", dact_from=a, b_kmajor=True, fallback=False)
            hip.gemm_rs(gx, W2, ga, b_kmajor=True, fallback=False)

        tc, ts = timeit(chain_b), timeit(separate_b)
        print(f"  its input gradients:                       chained {tc:6.1f} us ({fl / tc / 1e6:5.1f} TF) | three launches {ts:6.1f} us", flush=True)
