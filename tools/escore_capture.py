import os, sys, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_escore as T
from moleculesde_amd import hip
from moleculesde_amd import slabs  # noqa: E402
dev = torch.device("cuda", 0)
M_ = __import__("moleculesde_amd.geom3d.sde_2d_to_3d", fromlist=["x"]); M_.MOL_KERNEL_TRAIN = True
cpu_b, pl, ep, net, x, ea, basis = T._case(dev, int(sys.argv[1]) if len(sys.argv) > 1 else 16, 3)
net.train()
xd, ed, bd = x.to(dev).requires_grad_(True), ea.to(dev).requires_grad_(True), basis.to(dev)
w = torch.randn(ep.N, 3, device=dev)
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"
def step():
    net.zero_grad(set_to_none=True)
    xd.grad = None; ed.grad = None
    if mode == "batch":
        slabs.begin_param_grad_batch()
    out = net(ep, xd, ed, bd, pl)["gradient"]
    (out * w).sum().backward()
    if mode == "batch":
        slabs.finish_param_grad_batch()
for _ in range(3):
    step()
torch.cuda.synchronize()
print("eager ok", float(xd.grad.abs().sum()))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    if mode == "batch":
        slabs.new_param_grad_slot(dev)
    with torch.cuda.graph(g, stream=s):
        step()
print("captured")
g.replay(); torch.cuda.synchronize()
print("replayed", float(xd.grad.abs().sum()))
