"""GPU time of each part of the step (forward+backward), each captured alone in a hipGraph (GPU box)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import pretrain
from moleculesde_amd.geom3d import prepare_batch
from moleculesde_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
torch.manual_seed(0)
args = pretrain.readme_args()
tr = pretrain.Trainer(args, dev)
m = tr.models
b = prepare_batch(make_batch(256, seed=0), dev)
m["SDE_2Dto3D_model"].side_stream = None

def part_gin():
    h2 = m["model_2D"](b.x, b.edge_index, b.edge_attr)
    h2.square().mean().backward()
def part_schnet():
    _, h3 = m["model_3D"](b.x[:, 0], b.positions, b.batch, return_latent=True)
    h3.square().mean().backward()
h2d = m["model_2D"](b.x, b.edge_index, b.edge_attr).detach().requires_grad_(True)
h3d = m["model_3D"](b.x[:, 0], b.positions, b.batch, return_latent=True)[1].detach().requires_grad_(True)
def part_2d3d():
    m["SDE_2Dto3D_model"](h2d, b, anneal_power=0)["position"].backward()
def part_3d2d():
    lx, la = m["SDE_3Dto2D_model"](h3d, b, reduce_mean=True, continuous=True, train=True, anneal_power=0)
    (lx + la).backward()
def part_cl():
    l, _ = pretrain.dual_CL(h2d, h3d, args, tr.noise)
    l.backward()
def part_adam():
    tr.opt.step_from_grads()

for name, fn in (("GIN", part_gin), ("SchNet", part_schnet), ("2D->3D", part_2d3d), ("3D->2D head", part_3d2d), ("CL", part_cl), ("grad flatten + Adam", part_adam)):
    for _ in range(3):
        tr.opt.zero_grad(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    tr.opt.zero_grad()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{name:22s} fwd+bwd graph replay: {dt*1e3:7.3f} ms")
