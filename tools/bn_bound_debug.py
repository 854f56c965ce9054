import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for M, C, relu in ((684, 64, True), (684, 128, False), (3588, 300, True)):
    x = torch.randn(M, C, device=dev) * 2 + 0.5
    w = torch.randn(M, C, device=dev)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    res = []
    for bounded in (False, True):
        xx = x.clone().requires_grad_(True); gg = g.clone().requires_grad_(True); bb = b.clone().requires_grad_(True)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        cnt = torch.tensor([M], dtype=torch.int32, device=dev)
        ctx = hip.row_bounds({M: cnt}) if bounded else hip.row_bounds({})
        with ctx:
            y = hip.batch_norm_train(xx, gg, bb, rm, rv, 1e-5, 0.1, relu)
            (y * w).sum().backward()
        res.append((y.detach(), xx.grad, gg.grad, bb.grad, rm, rv))
    print(M, C, relu, [float((a - b_).abs().max()) for a, b_ in zip(*res)])
