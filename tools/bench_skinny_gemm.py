"""Launch time of the skinny node-level products of the dense head (3588 rows, <= 32 columns) through msde_gemm_ex."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip
import moleculesde_amd._lib as _L
dev = torch.device("cuda", 0)
for M, N, K, km in [(3588, 16, 364, True), (3588, 16, 512, True), (3588, 16, 128, True), (3588, 32, 128, False), (3588, 32, 300, False), (3588, 64, 364, True), (3588, 64, 128, False), (3588, 128, 300, False), (3588, 128, 128, False), (3588, 96, 512, True), (3588, 512, 16, False), (3588, 364, 128, True), (3588, 364, 119, False), (37136, 32, 300, False), (37136, 128, 32, False), (37136, 32, 128, False)]:
    A = torch.randn(M, K, device=dev)
    B = torch.randn(K, N, device=dev) if km else torch.randn(N, K, device=dev)
    C = torch.empty(M, N, device=dev)
    for _ in range(5):
        hip.gemm_ex(A, B, C, b_kmajor=km)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50):
            hip.gemm_ex(A, B, C, b_kmajor=km)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{M} x {N} x {K} b_kmajor={km}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch (back to back in a graph)")
