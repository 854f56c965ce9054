import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip
dev = torch.device("cuda", 0)
def run(M, N, K, S=0):
    torch.manual_seed(1)
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5
    out = torch.full((M, N), float("nan"), device=dev)
    hip.gemm_rs(A, W, out, t2=True, splits=S)
    ref = A.double() @ W.double().t()
    bad = ~((out.double() - ref).abs() < 1e-3)
    nb = int(bad.sum())
    msg = f"M={M} N={N} K={K} S={S}: bad {nb}"
    if nb:
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        msg += f" rows[{rows.numel()}] {rows[:6].tolist()}..{rows[-3:].tolist()} rowmod64 {sorted(set((rows % 64).tolist()))[:20]} cols[{cols.numel()}] {cols[:8].tolist()}..{cols[-3:].tolist()}"
        # is the error the contribution of some k range?  compare with partial sums
        r0, c0 = int(rows[0]), int(cols[0])
        diff = out[r0, c0].item() - ref[r0, c0].item()
        msg += f" | first bad ({r0},{c0}) out {out[r0, c0].item():.4g} ref {ref[r0, c0].item():.4g}"
        for k0 in range(0, K, 16):
            part = (A[r0, k0:k0 + 16].double() * W[c0, k0:k0 + 16].double()).sum().item()
            if abs(diff + part) < 1e-4:
                msg += f" == missing k[{k0}:{k0 + 16}]"
    print(msg, flush=True)
for K in (300, 332, 568, 600, 632, 664, 728, 32, 64, 96, 160):
    run(3588, 300, K)
for N in (300, 304, 320, 160, 128, 256):
    run(3588, N, 600, 4)
run(256 * 4, 300, 600, 4); run(640, 80, 96, 1)
