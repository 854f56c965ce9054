"""Host-side cost per launch (GPU box): eager torch op vs ctypes C-ABI call vs autograd.Function."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip, _lib
import torch.nn.functional as F
dev = torch.device("cuda", 0)
x = torch.randn(64, 32, device=dev); w = torch.randn(32, 32, device=dev); b = torch.randn(32, device=dev)
y = torch.empty(64, 32, device=dev)
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
print("F.linear (no grad)        issue/total us:", t(lambda: F.linear(x, w, b)))
print("torch add                 issue/total us:", t(lambda: x + x))
st = hip._stream()
print("raw ctypes msde_linear_fwd issue/total us:", t(lambda: _lib.call("msde_linear_fwd", hip._p(x), hip._p(w), hip._p(b), 64, 32, 32, hip._p(y), st)))
print("hip.linear (no grad)      issue/total us:", t(lambda: hip.linear(x, w, b)))
xg = x.clone().requires_grad_(True); wg = w.clone().requires_grad_(True)
print("hip.linear (grad fwd)     issue/total us:", t(lambda: hip.linear(xg, wg, b)))
def fb():
    hip.linear(xg, wg, b).sum().backward()
print("hip.linear fwd+bwd        issue/total us:", t(fb, 500))
def fb2():
    F.linear(xg, wg, b).sum().backward()
print("F.linear fwd+bwd          issue/total us:", t(fb2, 500))
