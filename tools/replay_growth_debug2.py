"""The in-suite failure of test_replay_after_workspaces_grew (nopair cross-check) outside pytest: the two tests that have to
run before it, then capture(small) -> step(big) -> replay(small), comparing the GRADIENTS of the first replay with the
all-eager sequence."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import moleculesde_amd.geom3d as G
from moleculesde_amd import hip
from moleculesde_amd.synthetic import make_batch
import test_gpu_models as T
dev = torch.device("cuda", 0)
for a in sys.argv[1:]:
    if "=" in a:
        mod, rest = a.split(":"); attr, val = rest.split("=")
        setattr(__import__(mod, fromlist=["x"]), attr, eval(val))
hip.CFCONV_PAIR = False
if "novariants" not in sys.argv:
    T.test_golden_f3_variants_2d3d(dev, "m01")
    T.test_golden_f3_variants_2d3d(dev, "vp")
if "nofar" not in sys.argv:
    T.test_eager_steps_with_gpu_far_behind_host(dev)


class Fixed(G.DeviceNoise):
    def __init__(self):
        super().__init__(seed=3)
        g = torch.Generator().manual_seed(9)
        self.big, self.ints = torch.randn(4096, 3, generator=g).to(dev), torch.randint(0, 1000, (512,), generator=g).to(dev)
        self.perm = {}

    def randn_like(self, x): return self.big[:x.size(0)].clone()
    def randint(self, high, size, device): return self.ints[:size[0]].clone()

    def randperm_pair(self, n, device):
        if n not in self.perm:
            self.perm[n] = (torch.randperm(n, generator=torch.Generator().manual_seed(n)).int().to(device),
                            torch.randperm(n, generator=torch.Generator().manual_seed(n + 1)).int().to(device))
        return self.perm[n]


from moleculesde_amd import slabs
_retire0 = slabs._retire


def _retire_logged(bufs):
    import traceback
    fr = traceback.extract_stack(limit=3)[0]
    print("   _retire:", [None if b is None else (tuple(b.shape), hex(b.data_ptr())) for b in bufs], "from", fr.name, fr.lineno, "captured", slabs._CAPTURED, flush=True)
    _retire0(bufs)


if "logretire" in sys.argv:
    slabs._retire = _retire_logged        # (hip.py's own callers go through its imported name: patch both)
    hip._retire = _retire_logged
small = G.prepare_batch(make_batch(8, seed=81), dev)
big = G.prepare_batch(make_batch(96, seed=82), dev)
res = []
for use_graph in (True, False):
    tr = T._quiet_trainer(dev, seed=6)
    for m in tr.models.values():
        T.disable_dropout(m)
    tr.noise = Fixed()
    tr.models["SDE_2Dto3D_model"].noise = tr.noise
    names = [f"{mn}.{pn}" for mn, m in tr.models.items() for pn, p in m.named_parameters()]
    params = [p for m in tr.models.values() for p in m.parameters()]
    print("run graph" if use_graph else "run eager", flush=True)
    tr.step(small)
    if use_graph:
        print("  capture", flush=True)
        tr.capture(small)
    print("  step(big)", flush=True)
    tr.step(big)
    print("  first small step behind it", flush=True)
    if "dirty" in sys.argv:        # whatever the eager step handed back to the caching allocator now holds NaN
        torch.cuda.synchronize()
        xs = [torch.full((n,), float("nan"), device=dev) for n in (1 << 8, 1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24) for _ in range(24)]
        del xs
        torch.cuda.synchronize()
    if "refresh" in sys.argv and use_graph:
        hip.refresh_weight_t()
    (tr.step_graph if use_graph else tr.step)(small)
    torch.cuda.synchronize()
    res.append(([None if p.grad is None else p.grad.detach().clone() for p in params], tr.opt.flat_p.clone(), names))
(g0, p0, names), (g1, p1, _) = res
print("parameters after the first small step behind the big one: graph/eager rel diff", float((p0 - p1).norm() / p1.norm()))
bad, off = [], 0
params_ = [pp for m in tr.models.values() for pp in m.parameters()]
order = {id(pp): n for n, pp in zip(names, params_)}
for pp in tr.opt.params:
    n = pp.numel()
    a, b = p0[off:off + n], p1[off:off + n]
    d = float((a - b).norm() / (b.norm() + 1e-30))
    if d > 1e-7:
        bad.append((d, order.get(id(pp), "?"), n))
    off += n
print(len(bad), "of", len(tr.opt.params), "parameters differ after ONE replay; in optimiser order:", [(round(d, 6), n) for d, n, _ in bad][:60])
