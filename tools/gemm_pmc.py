"""gemm_ex 3588x728x728 with and without the global->LDS staging (debug flag 256), 20 launches each, for
rocprofv3 --pmc SQ_* (which wait bucket do the 15 us of staging land in?)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip
dev = torch.device("cuda", 0)
M, N, K = 3588, 728, 728
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / 27; b = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev)
for flags in (0, 256):
    for _ in range(20):
        hip.gemm_ex(A, W, out, bias=b, act="silu", _debug_flags=flags)
    torch.cuda.synchronize()
    # marker launch between the two groups
    torch.zeros(1, device=dev).add_(1)
    torch.cuda.synchronize()
