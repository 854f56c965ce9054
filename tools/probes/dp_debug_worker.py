"""Worker of tests/test_gpu_dp.py: one of 2 ranks that share cuda:0 (gloo backend on device tensors).

Checks the data-parallel Trainer step (examples/pretrain_MoleculeSDE.py:154-156,331-337 is the
single-process step it generalises), eager and hipGraph forms:
  A. both ranks see the SAME batch and the same noise  -> parameters after 3 steps == a 1-rank run;
  B. ranks see DIFFERENT shards                        -> parameters == a 1-rank run that applies Adam to the
     mean of the two shards' gradients (BatchNorm statistics per shard, as under DP).
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import disable_dropout  # noqa: E402
from moleculesde_amd import dp, pretrain  # noqa: E402
import moleculesde_amd.geom3d as G  # noqa: E402
from moleculesde_amd.synthetic import make_batch  # noqa: E402

os.environ["MSDE_DP_BACKEND"] = "gloo"
rank, world, local = dp.init_from_env("cuda")
assert world == 2 and dist.get_backend() == "gloo"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
FULL = len(sys.argv) > 2 and sys.argv[2] == "full"
args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=1 if FULL else 0, emb_dim=64)


class FixedNoise(G.DeviceNoise):
    """Same draws on every call and every rank (dropout is disabled too): runs are comparable bit for bit."""

    def __init__(self, seed):
        self.g0, self.cache = seed, {}

    def _get(self, key, make, device):
        # device copies are made once (first eager call) and cloned afterwards: no H2D copy inside a graph capture
        if key not in self.cache:
            import zlib
            seed = self.g0 * 1000003 + zlib.crc32(repr(key).encode())     # a function of the KEY, not of call order
            self.cache[key] = make(torch.Generator().manual_seed(seed)).to(device)
        return self.cache[key].clone()

    def randn_like(self, x):
        return self._get(("n", tuple(x.shape)), lambda g: torch.randn(x.shape, generator=g), x.device)

    def randint(self, high, size, device):
        return self._get(("i", high, tuple(size)), lambda g: torch.randint(0, high, size, generator=g), device)

    def randperm(self, n, device):
        return self._get(("p", n), lambda g: torch.randperm(n, generator=g), device)

    def rand(self, n, device):
        return self._get(("r", n), lambda g: torch.rand(n, generator=g), device)


def trainer(seed, dp_on):
    torch.manual_seed(seed)
    tr = pretrain.Trainer(args, dev)
    tr.dp_enabled = dp_on
    for m in tr.models.values():
        disable_dropout(m)
    tr.noise = FixedNoise(5)
    for k in ("SDE_2Dto3D_model", "SDE_3Dto2D_model"):
        if k in tr.models:
            tr.models[k].noise = tr.noise
    return tr


def same_params(a, b):
    for k in a.models:
        b.models[k].load_state_dict(a.models[k].state_dict())


def rel(a, b):
    return float((a - b).norm() / b.norm())


shards = [G.prepare_batch(make_batch(24, seed=31 + r), dev) for r in range(2)]


names = None
for use_graph in (True,):
    tr_dp, tr_1 = trainer(4, True), trainer(4, False)
    same_params(tr_dp, tr_1)
    names = {id(p): f"{k}.{n}" for k, m in tr_dp.models.items() for n, p in m.named_parameters()}
    tr_dp.step(shards[rank])
    tr_dp.capture(shards[rank])
    same_params(tr_1, tr_dp)
    tr_dp.opt.m.zero_(); tr_dp.opt.v.zero_(); tr_dp.opt.step_dev.zero_()
    for it in range(3):
        g, held, with_adam = tr_dp._graphs[id(shards[rank])]
        g.replay()
        torch.cuda.synchronize()
        local = tr_dp.opt.flat_g.clone()
        # eager gradient of MY shard with the reference trainer (same parameters)
        loss, _ = tr_1.losses(shards[rank])
        tr_1.opt.zero_grad()
        tr_1._backward(loss)
        mine = tr_1.opt.gather_grads().clone()
        torch.cuda.synchronize()
        worst = []
        for p, o, sz in zip(tr_dp.opt.params, tr_dp.opt.offsets, tr_dp.opt.sizes):
            a, c = local[o:o + sz].double(), mine[o:o + sz].double()
            worst.append((float((a - c).norm()) / max(float(c.norm()), 1e-12), float(c.norm()), names[id(p)]))
        worst.sort(reverse=True)
        print(f"rank {rank} it {it}: graph-vs-eager local gradient, worst:", [(f"{w[0]:.2e}", f"{w[1]:.2e}", w[2]) for w in worst[:5]], flush=True)
        loss, _ = tr_dp.losses(shards[rank])
        tr_dp.opt.zero_grad()
        tr_dp._backward(loss)
        own = tr_dp.opt.gather_grads().clone()
        torch.cuda.synchronize()
        eps_idx = [(o, names[id(p)]) for p, o, sz in zip(tr_dp.opt.params, tr_dp.opt.offsets, tr_dp.opt.sizes) if names[id(p)].endswith(".eps")]
        print(f"rank {rank} it {it}: eps grads (dp graph | dp eager | ref eager):", [(n.split(".")[2], f"{float(local[o]):.5e}", f"{float(own[o]):.5e}", f"{float(mine[o]):.5e}") for o, n in eps_idx], flush=True)
        print(f"rank {rank} it {it}: max|graph-own| {float((local-own).abs().max()):.3e} max|own-ref| {float((own-mine).abs().max()):.3e}", flush=True)
        big = []
        for p, o, sz in zip(tr_dp.opt.params, tr_dp.opt.offsets, tr_dp.opt.sizes):
            a, c = local[o:o + sz].double(), own[o:o + sz].double()
            r_ = float((a - c).norm()) / max(float(c.norm()), 1e-12)
            if r_ > 1e-4 and float(c.norm()) > 1e-4:
                big.append((f"{r_:.2e}", f"{float(c.norm()):.2e}", names[id(p)]))
        print(f"rank {rank} it {it}: {len(big)} parameters with rel > 1e-4 and |g| > 1e-4:", big[:40], flush=True)
        tr_dp.opt.flat_g.copy_(local)
        print(f"rank {rank} it {it}: params equal before step: {rel(tr_dp.opt.flat_p, tr_1.opt.flat_p):.3e}", flush=True)
        tr_dp._allreduce_and_adam()
        gs = []
        for sh in shards:
            loss, _ = tr_1.losses(sh)
            tr_1.opt.zero_grad()
            tr_1._backward(loss)
            gs.append(tr_1.opt.gather_grads().clone())
        tr_1.opt.flat_g.copy_((gs[0] + gs[1]) * 0.5)
        tr_1.opt.step()
        torch.cuda.synchronize()
        print(f"rank {rank} it {it}: params after step: {rel(tr_dp.opt.flat_p, tr_1.opt.flat_p):.3e}", flush=True)
dp.barrier()
dist.destroy_process_group()
