"""Per-parameter difference between the gradients of a captured step (graph without Adam, the data-parallel structure)
and of the same step launched from the host, at identical parameters and noise."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import moleculesde_amd.geom3d as G
from moleculesde_amd import pretrain
from moleculesde_amd.synthetic import make_batch
from helpers import disable_dropout

dev = torch.device("cuda", 0)
args = pretrain.readme_args(emb_dim=64, SDE_coeff_generative_3Dto2D=0)


class FixedNoise(G.DeviceNoise):
    def __init__(self, seed):
        super().__init__(seed=seed)
        g = torch.Generator().manual_seed(seed)
        self.big = torch.randn(8192, 3, generator=g).to(dev)
        self.ints = torch.randint(0, 1000, (4096,), generator=g).to(dev)

    def randn_like(self, x):
        return self.big[:x.size(0)].clone()

    def randint(self, high, size, device):
        return self.ints[:size[0]].clone()

    def randperm_pair(self, n, device):
        self.calls = 0
        return super().randperm_pair(n, device)


torch.manual_seed(4)
tr = pretrain.Trainer(args, dev)
tr.adam_outside_graph = True
for m in tr.models.values():
    disable_dropout(m)
tr.noise = FixedNoise(5)
tr.models["SDE_2Dto3D_model"].noise = tr.noise
b = G.prepare_batch(make_batch(24, seed=31), dev)
tr.step(b)
tr.capture(b)
names = {id(p): f"{k}.{n}" for k, m in tr.models.items() for n, p in m.named_parameters()}
for it in range(2):
    g, held, with_adam = tr._graphs[id(b)]
    g.replay()
    torch.cuda.synchronize()
    gg = tr.opt.flat_g.clone()
    loss, _ = tr.losses(b)
    tr.opt.zero_grad()
    tr._backward(loss)
    ge = tr.opt.gather_grads().clone()
    torch.cuda.synchronize()
    worst = []
    for p, o, sz in zip(tr.opt.params, tr.opt.offsets, tr.opt.sizes):
        a, c = gg[o:o + sz].double(), ge[o:o + sz].double()
        worst.append((float((a - c).norm()) / max(float(c.norm()), 1e-12), float(c.norm()), names[id(p)]))
    worst.sort(reverse=True)
    print("iteration", it, "graph loss", float(tr._graph_loss[id(b)]), "eager loss", float(loss))
    for w in worst[:12]:
        print("   rel %.3e  |g| %.3e  %s" % w)
    tr.opt.step()
    tr._refresh_weights()
