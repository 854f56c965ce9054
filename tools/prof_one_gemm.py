import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip
hip.set_linear_mode("hip")
dev = torch.device("cuda", 0)
M, N, K = 3588, 600, 300
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
for _ in range(20):
    hip.linear(x, w, b)
torch.cuda.synchronize()
M, N, K = 49090, 128, 128
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
for _ in range(20):
    hip.linear(x, w, b)
torch.cuda.synchronize()
