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
from moleculesde_amd import wcache  # noqa: E402

os.environ["MSDE_DP_BACKEND"] = "gloo"
rank, world, local = dp.init_from_env("cuda")
assert world == 2 and dist.get_backend() == "gloo"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
FULL = len(sys.argv) > 2 and sys.argv[2] == "full"
if len(sys.argv) > 2 and sys.argv[2] == "overlap":      # the bucket-wise tail (pretrain.DP_OVERLAP): off by default, kept green
    pretrain.DP_OVERLAP = True
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

for use_graph in (False, True):
    # ---- A: identical data on both ranks ----------------------------------------------------------------------
    tr_dp, tr_1 = trainer(3, True), trainer(3, False)
    same_params(tr_dp, tr_1)
    b = shards[0]
    if use_graph:
        tr_dp.step(b); tr_1.step(b)
        tr_dp.capture(b); tr_1.capture(b)
    for _ in range(3):
        if use_graph:
            tr_dp.step_graph(b); tr_1.step_graph(b)
        else:
            tr_dp.step(b); tr_1.step(b)
    torch.cuda.synchronize()
    d = rel(tr_dp.opt.flat_p, tr_1.opt.flat_p)
    print(f"rank {rank} graph={use_graph} identical shards: distance {d:.2e}", flush=True)
    assert d < 1e-6, ("identical shards", use_graph, d)
    # every rank holds the same parameters
    mine = tr_dp.opt.flat_p.clone()
    other = mine.clone()
    dist.broadcast(other, src=0)
    assert torch.equal(mine, other), "replicas diverged"

    # ---- B: different shards --------------------------------------------------------------------------------
    tr_dp, tr_1 = trainer(4, True), trainer(4, False)
    same_params(tr_dp, tr_1)
    if use_graph:
        tr_dp.step(shards[rank])
        tr_dp.capture(shards[rank])
        # the warm-up step moved tr_dp: both restart from the reference's point.  This rewrites tr_dp's parameters from
        # OUTSIDE the optimiser after the capture: the replay must still see them (wcache.sync_weight_copies in step_graph
        # refreshes the cached re-laid-out weight copies the forward products read)
        same_params(tr_1, tr_dp)
        tr_dp.opt.m.zero_(); tr_dp.opt.v.zero_(); tr_dp.opt.step_dev.zero_()
    for _ in range(3):
        if use_graph:
            tr_dp.step_graph(shards[rank])
        else:
            tr_dp.step(shards[rank])
        # 1-rank reference: gradient of each shard, mean, Adam
        gs = []
        for sh in shards:
            loss, _ = tr_1.losses(sh)
            tr_1.opt.zero_grad()
            tr_1._backward(loss)
            gs.append(tr_1.opt.gather_grads().clone())
        tr_1.opt.flat_g.copy_((gs[0] + gs[1]) * 0.5)
        tr_1.opt.step()
    torch.cuda.synchronize()
    d = rel(tr_dp.opt.flat_p, tr_1.opt.flat_p)
    assert d < 2e-5, ("different shards", use_graph, d)
    print(f"rank {rank} graph={use_graph} OK (mean-gradient distance {d:.2e})", flush=True)

dp.barrier()
dist.destroy_process_group()
sys.stdout.write(f"RANK{rank}DONE\n")       # ONE write: the two ranks share a pipe, separate writes interleave
sys.stdout.flush()
