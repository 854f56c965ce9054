"""Is msde_gemm_t2b (and msde_gemm_t2, the control) bitwise stable while other kernels share the chip?  Each product is
computed alone once, then repeated while a second stream runs unrelated work; any differing repetition is printed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moleculesde_amd import hip

dev = torch.device("cuda", 0)
shapes = [(550, 64, 64), (560, 64, 128), (550, 128, 64), (550, 64, 16), (550, 300, 64), (3588, 300, 300), (3588, 128, 300),
          (550, 32, 64), (1100, 64, 64), (550, 192, 64), (600, 64, 192)]
side = torch.cuda.Stream()
big = torch.randn(4096, 4096, device=dev)
junk = torch.empty(64 << 20, device=dev)
for (M, N, K) in shapes:
    if not hip.t2_ok(M, N, K):
        print("skip", M, N, K); continue
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev)
    W = torch.nn.Parameter((torch.randn(N, K, generator=g) / K ** 0.5).to(dev))
    b = torch.randn(N, generator=g).to(dev)
    planes, ld = hip.weight_planes(W, False)
    for name, kw in (("t2b", dict(t2b_ld=ld)), ("t2", dict(t2=True))):
        B = planes if name == "t2b" else W.detach()
        ref = torch.empty(M, N, device=dev)
        hip.gemm_rs(A, B, ref, bias=b, act="ssp", N=N, K=K, **kw)
        torch.cuda.synchronize()
        bad = 0
        outs = [torch.empty(M, N, device=dev) for _ in range(8)]
        for rep in range(60):
            with torch.cuda.stream(side):
                for _ in range(3):
                    junk.fill_(float(rep))
                    torch.mm(big, big)
            for o in outs:
                o.fill_(float("nan"))
                hip.gemm_rs(A, B, o, bias=b, act="ssp", N=N, K=K, **kw)
            torch.cuda.synchronize()
            for o in outs:
                if not torch.equal(o, ref):
                    bad += 1
                    if bad <= 3:
                        df = (o - ref).abs()
                        print("   ", name, (M, N, K), "rep", rep, "differs: max", float(df.max()), "count", int((df > 0).sum()),
                              "nan", int(torch.isnan(o).sum()))
        print(name, (M, N, K), "bad repetitions:", bad, "of", 60 * 8, flush=True)
