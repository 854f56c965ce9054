"""Does msde_gemm_t2b (or msde_gemm_t2) write outside its output?  Output = a slice in the middle of a sentinel-filled buffer."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moleculesde_amd import hip
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for (M, N, K) in [(3442, 128, 64), (3442, 32, 64), (3442, 64, 128), (3442, 64, 32), (550, 64, 64), (3588, 300, 300), (777, 80, 64)]:
    A = torch.randn(M, K, device=dev)
    W = torch.nn.Parameter(torch.randn(N, K, device=dev) / K ** 0.5)
    b = torch.randn(N, device=dev)
    planes, ld = hip.weight_planes(W, False)
    for kind in ("t2b", "t2"):
        G = 1 << 16
        buf = torch.full((2 * G + M * N,), 12345.0, device=dev)
        out = buf[G:G + M * N].view(M, N)
        inbuf = torch.full((2 * G + M * K,), 777.0, device=dev)
        Ain = inbuf[G:G + M * K].view(M, K); Ain.copy_(A)
        if kind == "t2b":
            hip.gemm_rs(Ain, planes, out, bias=b, N=N, K=K, t2b_ld=ld)
        else:
            hip.gemm_rs(Ain, W.detach(), out, bias=b, N=N, K=K, t2=True)
        torch.cuda.synchronize()
        lo, hi = buf[:G], buf[G + M * N:]
        ok = bool((lo == 12345.0).all()) and bool((hi == 12345.0).all()) and bool((out != 12345.0).all())
        ok_in = bool((inbuf[:G] == 777.0).all()) and bool((inbuf[G + M * K:] == 777.0).all()) and torch.equal(Ain, A)
        ref = A.double() @ W.detach().double().t() + b.double()
        print(kind, (M, N, K), "guards intact:", ok, "input intact:", ok_in, "max err %.2g" % float((out.double() - ref).abs().max()),
              "" if ok else "front %d back %d" % (int((lo != 12345.0).sum()), int((hi != 12345.0).sum())), flush=True)
