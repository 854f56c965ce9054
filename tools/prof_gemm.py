"""Launch each GEMM shape of the step a fixed number of times (for rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip
import torch.nn.functional as F
dev = torch.device("cuda", 0)
SHAPES = [(3588, 600, 300), (3588, 300, 600), (3588, 128, 300), (3588, 300, 300), (49090, 128, 51), (49090, 128, 128),
          (35186, 32, 300), (35186, 128, 64), (35186, 32, 32)]
for M, N, K in SHAPES:
    x = torch.randn(M, K, device=dev, requires_grad=True); w = torch.randn(N, K, device=dev, requires_grad=True)
    b = torch.randn(N, device=dev, requires_grad=True); gy = torch.randn(M, N, device=dev)
    for _ in range(10):
        y = hip.linear(x, w, b); y.backward(gy)
    torch.cuda.synchronize()
    # marker kernel between shapes: a fill of M*N+K elements
    torch.zeros(M * 7 + N * 3 + K, device=dev)
    for _ in range(10):
        y = F.linear(x, w, b); y.backward(gy)
    torch.cuda.synchronize()
