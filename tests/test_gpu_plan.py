"""Batch construction on the GPU (csrc/plan.hip, moleculesde_amd/bucket.py) and capacity buckets.

1. The device-built plan is BIT-IDENTICAL to the host plan of moleculesde_amd/plan.py (whose extend_graph is checked
   against the reference's algorithm in tests/test_host_logic.py) on every valid entry, for PCQM4Mv2-shaped batches and
   edge cases (1-atom / 2-atom molecules, rings, maximum size 32, a molecule without bonds).
2. A pretrain step on a padded bucket (row bounds) reproduces the step on the exact-size batch: losses and gradients.
3. One captured hipGraph serves different batches.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import assert_close, disable_dropout  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from moleculesde_amd import _lib
    _lib.load()
    return torch.device("cuda", 0)


def _edge_case_batch():
    from moleculesde_amd.batch import Batch
    from moleculesde_amd.synthetic import make_molecule
    rng = np.random.default_rng(5)
    mols = [make_molecule(rng, n) for n in (2, 3, 32, 1, 20, 7, 32, 2)]
    return Batch.from_data_list(mols)


def _check_plan_equal(bk, cpu_b):
    from moleculesde_amd import plan as P
    hp = P.build_plan(cpu_b.clone())
    P.dense_plan(hp, cpu_b)
    pl = bk.plan
    torch.cuda.synchronize()
    ok, sizes = bk.check()
    assert ok, sizes
    N, Eb, Ee = hp.N, hp.bond.E, hp.ext.E
    assert sizes["N"] == N and sizes["E_b"] == Eb and sizes["E_e"] == Ee and sizes["P"] == hp.dense.P
    assert sizes["n_max"] == hp.N_max and sizes["E_r_bound"] == hp.E_r_cap
    eq = lambda a, b, what: (torch.equal(a.cpu(), b.cpu()), what)
    checks = [
        eq(pl.mol_ptr, hp.mol_ptr, "mol_ptr"), eq(pl.batch_i32[:N], hp.batch_i32, "batch"),
        eq(pl.atom_codes[:N], hp.atom_codes, "atom_codes"), eq(pl.z_codes[:N], hp.z_codes, "z_codes"),
        eq(pl.atom_list_ptr, hp.atom_list_ptr, "atom_list_ptr"),
        eq(pl.atom_list_nodes[:N * 9], hp.atom_list_nodes, "atom_list_nodes"),
        eq(bk.pair_ptr, hp.dense.pair_ptr, "pair_ptr"),
    ]
    zp, zn = P.z_lists(hp, 119)
    checks += [eq(pl.z_list[1], zp, "z_list_ptr"), eq(pl.z_list[2][:N], zn, "z_list_nodes")]
    for name, E in (("bond", Eb), ("ext", Ee)):
        a, h = getattr(pl, name), getattr(hp, name)
        checks += [eq(a.rowptr[:N + 1], h.rowptr, name + ".rowptr"), eq(a.src[:E], h.src, name + ".src"),
                   eq(a.dst[:E], h.dst, name + ".dst"), eq(a.rowptr_s[:N + 1], h.rowptr_s, name + ".rowptr_s"),
                   eq(a.perm_s[:E], h.perm_s, name + ".perm_s")]
        # padded tails
        assert bool((a.rowptr[N:] == E).all()) and bool((a.src[E:] == -1).all()) and bool((a.dst[E:] == -1).all())
    checks += [eq(pl.bond_codes[:Eb], hp.bond_codes, "bond_codes"), eq(pl.bond_type[:Eb], hp.bond_type, "bond_type")]
    bad = [w for ok_, w in checks if not ok_]
    assert not bad, bad
    assert bool((pl.batch_i32[N:] == pl.B).all())
    # the extended graph itself: the same edge SET as the loader's extended_edge_index, in canonical order
    ext = torch.stack([pl.ext.src[:Ee], pl.ext.dst[:Ee]]).cpu().long()
    ref = cpu_b.extended_edge_index
    key = lambda e: set(map(tuple, e.t().tolist()))
    assert key(ext) == key(ref)


@pytest.mark.parametrize("which", ["pcqm256", "edge_cases", "tight_caps"])
def test_device_plan_bit_identical_to_host_plan(dev, which):
    from moleculesde_amd import bucket as BK
    from moleculesde_amd.synthetic import make_batch
    cpu_b = make_batch(256, seed=17) if which != "edge_cases" else _edge_case_batch()
    need = BK.raw_sizes(cpu_b)
    caps = BK.Caps.covering([need], n_max=32)
    if which == "pcqm256":          # generous capacities: long padded tails
        caps = BK.Caps(need["B"], need["N"] + 700, need["E_b"] + 900, need["E_e"] + 5000, need["E_r"] + 3000,
                       need["P"] + 4000, 32)
    bk = BK.Bucket(caps, dev)
    for rep in range(2):            # second load over stale contents
        bk.load(BK.pack_raw(cpu_b, caps))
        bk.build_plan_on_device()
        _check_plan_equal(bk, cpu_b)
        if rep == 0:                # a different batch in between leaves no residue
            other = make_batch(need["B"], seed=99, sizes=[3] * need["B"])
            bk.load(BK.pack_raw(other, caps))
            bk.build_plan_on_device()
            _check_plan_equal(bk, other)


def _fixed_noise(G, dev):
    class FixedNoise(G.DeviceNoise):
        """Values are a function of (kind, ATOM / MOLECULE index) only -- never of the padded shape."""

        def randn_like(self, x):
            g = torch.Generator().manual_seed(7 + x.dim())
            if x.dim() == 3:        # dense-head noise [B, Nm, Nm | classes]: independent of the padded Nm
                last = 32 if x.size(2) == x.size(1) else x.size(2)
                big = torch.randn(256, 32, last, generator=g)
                return big[:x.size(0), :x.size(1), :x.size(2)].contiguous().to(x.device)
            big = torch.randn((8192,) + tuple(x.shape[1:]), generator=g)
            return big[:x.size(0)].to(x.device)

        def randint(self, high, size, device):
            return torch.randint(0, high, (4096,), generator=torch.Generator().manual_seed(3))[:size[0]].to(device)
    return FixedNoise()


@pytest.mark.parametrize("full", [False, True])
def test_bucket_step_matches_exact_batch(dev, full):
    """Losses and every parameter gradient of one pretrain step: padded bucket + row bounds + device-built plan vs the
    exact-size batch with the host plan.  Contrastive permutation: identity on both sides is impossible (it is drawn
    on the device), so the contrastive negatives come from a fixed permutation of the VALID atoms."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import bucket as BK, hip, pretrain
    from moleculesde_amd.synthetic import make_batch
    args = pretrain.readme_args(emb_dim=64, SDE_coeff_generative_3Dto2D=1 if full else 0)
    torch.manual_seed(21)
    tr = pretrain.Trainer(args, dev)
    tr.overlap_streams = False
    for m in tr.models.values():
        disable_dropout(m)
    cpu_b = make_batch(48, seed=23)
    need = BK.raw_sizes(cpu_b)
    N = need["N"]
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(1))

    def run(batch, n_rows):
        noise = _fixed_noise(G, dev)
        p1 = torch.arange(n_rows)
        p1[:N] = perm
        p2 = torch.arange(n_rows)
        p2[:N] = perm.flip(0)
        noise.randperm_pair = lambda n, device: (p1.to(device).int(), p2.to(device).int())
        tr.noise = noise
        for k in ("SDE_2Dto3D_model", "SDE_3Dto2D_model"):
            if k in tr.models:
                tr.models[k].noise = noise
        tr.opt.zero_grad()
        loss, parts = tr.losses(batch)
        tr._backward(loss)
        torch.cuda.synchronize()
        return ({k: float(v) for k, v in parts.items()}, tr.opt.gather_grads().clone(),
                {k: {n: v.clone() for n, v in mm.state_dict().items()} for k, mm in tr.models.items()})

    sd0 = {k: {n: v.clone() for n, v in m.state_dict().items()} for k, m in tr.models.items()}
    exact = G.prepare_batch(cpu_b.clone(), dev)
    hip.clear_row_bounds()
    parts_e, g_e, _ = run(exact, N)
    for k, m in tr.models.items():
        m.load_state_dict(sd0[k])                       # BatchNorm running statistics back to the start
    caps = BK.Caps(need["B"], N + 300, need["E_b"] + 600, need["E_e"] + 3000, need["E_r"] + 2500, need["P"] + 3000, 24)
    bk = BK.Bucket(caps, dev)
    bk.load(BK.pack_raw(cpu_b, caps))
    bk.build_plan_on_device()
    bk.activate()
    try:
        parts_b, g_b, sd_b = run(bk.batch, caps.N)
    finally:
        hip.clear_row_bounds()
    for k in parts_e:
        assert abs(parts_b[k] - parts_e[k]) <= 2e-5 * abs(parts_e[k]) + 1e-7, (k, parts_b[k], parts_e[k])
    # per-parameter relative L2 error (different row capacities give different split geometries in the BatchNorm /
    # weight-gradient reductions, i.e. a different fp32 summation order -- not a different result)
    names = {id(p): f"{k}.{n}" for k, m in tr.models.items() for n, p in m.named_parameters()}
    nmax = max(float(g_e[o:o + sz].norm()) for o, sz in zip(tr.opt.offsets, tr.opt.sizes))
    worst = []
    for p, o, sz in zip(tr.opt.params, tr.opt.offsets, tr.opt.sizes):
        a, b = g_b[o:o + sz].double(), g_e[o:o + sz].double()
        den = max(float(b.norm()), 1e-3 * nmax)
        worst.append((float((a - b).norm()) / den, names[id(p)]))
    worst.sort(reverse=True)
    print("bucket vs exact, worst per-parameter rel-L2 gradient errors:", worst[:6])
    assert worst[0][0] < 2e-3, worst[:10]
    # BatchNorm running statistics saw the valid rows only
    for k, m in tr.models.items():
        m.load_state_dict(sd0[k])
    _, _, _ = run(exact, N)
    for k, m in tr.models.items():
        for n, v in m.state_dict().items():
            if "running_" in n:
                assert_close(sd_b[k][n], v, 1e-5, 1e-6, f"{k}.{n}")


def test_one_graph_serves_different_batches(dev):
    """ONE captured hipGraph (device-side plan construction + forward + backward + Adam) replayed on three DIFFERENT
    batches, each compared with the eager step on the exact-size batch from the same parameters / optimiser state /
    random streams: same loss terms, same updated parameters."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import bucket as BK, hip, pretrain
    from moleculesde_amd.synthetic import make_batch
    args = pretrain.readme_args(emb_dim=64)
    torch.manual_seed(31)
    tr = pretrain.Trainer(args, dev)
    tr.overlap_streams = False
    for m in tr.models.values():
        disable_dropout(m)

    class Noise(G.DeviceNoise):
        """Position noise / time steps from fixed device tensors (padding independent, capturable); the contrastive
        permutation and the dense-head noise stay on the device generators (functions of seed, counter, index)."""

        def __init__(self):
            super().__init__(seed=77)
            g = torch.Generator().manual_seed(5)
            self.big = torch.randn(8192, 3, generator=g).to(dev)
            self.ints = torch.randint(0, 1000, (4096,), generator=g).to(dev)

        def randn_like(self, x):
            assert x.dim() == 2 and x.size(1) == 3
            return self.big[:x.size(0)].clone()

        def randint(self, high, size, device):
            return self.ints[:size[0]].clone()
    noise = Noise()
    # the dense head must see a DeviceNoise (in-kernel draws), the 2D->3D model the fixed tensors
    tr.noise = noise
    tr.models["SDE_2Dto3D_model"].noise = noise
    head_noise = G.DeviceNoise(seed=99)
    tr.models["SDE_3Dto2D_model"].noise = head_noise

    cpu = [make_batch(40, seed=s) for s in (51, 52, 53)]
    needs = [BK.raw_sizes(b) for b in cpu]
    caps = BK.Caps.covering(needs, n_max=24)
    bk = tr.make_bucket(caps)
    blobs = [BK.pack_raw(b, caps).to(dev) for b in cpu]
    tr.capture_bucket(bk, blobs[0])
    snap = lambda: (tr.opt.flat_p.clone(), tr.opt.m.clone(), tr.opt.v.clone(), tr.opt.step_dev.clone(),
                    tr.step_counter.clone(), {k: {n: v.clone() for n, v in m.state_dict().items()} for k, m in tr.models.items()})

    def restore(s):
        tr.opt.flat_p.copy_(s[0]); tr.opt.m.copy_(s[1]); tr.opt.v.copy_(s[2]); tr.opt.step_dev.copy_(s[3])
        tr.step_counter.copy_(s[4])
        for k, m in tr.models.items():
            m.load_state_dict(s[5][k])
    calls = (noise.calls, head_noise.calls)
    for i in (1, 2, 0, 1):
        s0 = snap()
        for k in tr.log:
            tr.log[k].zero_()
        tr.step_bucket(bk, blobs[i])
        torch.cuda.synchronize()
        ok, sizes = bk.check()
        assert ok and sizes["N"] == needs[i]["N"] and sizes["E_e"] == needs[i]["E_e"], sizes
        got_log = {k: float(v) for k, v in tr.log.items()}
        got_p = tr.opt.flat_p.clone()
        restore(s0)
        # eager step on the exact-size batch with the host plan; host-side draw counters as they were at capture time
        noise.calls, head_noise.calls = calls[0] - 1, calls[1] - 1
        for k in tr.log:
            tr.log[k].zero_()
        exact = G.prepare_batch(cpu[i].clone(), dev)
        tr.step(exact)
        torch.cuda.synchronize()
        for k in got_log:
            ref = float(tr.log[k])
            assert abs(got_log[k] - ref) <= 5e-5 * abs(ref) + 1e-7, (i, k, got_log[k], ref)
        # the two parameter UPDATES point the same way.  (Adam turns every gradient into a step of ~lr: parameters whose
        # gradient is pure rounding noise -- biases in front of a BatchNorm, key biases under a softmax -- move by +-lr
        # with a sign that depends on the summation order, so the comparison is a cosine, not an element-wise one; the
        # gradients themselves are compared element-wise in test_bucket_step_matches_exact_batch.)
        ua, ub = (got_p - s0[0]).double(), (tr.opt.flat_p - s0[0]).double()
        cos = float((ua * ub).sum() / (ua.norm() * ub.norm()))
        assert cos > 0.995, (i, cos)
        hip.clear_row_bounds()


def test_blob_feeder_prefetch(dev):
    """bucket.BlobFeeder: pinned blobs staged one step ahead on a copy stream arrive in the bucket intact and in order."""
    from moleculesde_amd import bucket as BK
    from moleculesde_amd.synthetic import make_batch
    cpu = [make_batch(8, seed=60 + s) for s in range(5)]
    caps = BK.Caps.covering([BK.raw_sizes(b) for b in cpu])
    bk = BK.Bucket(caps, dev)
    pinned = [BK.pack_raw(b, caps, pin=True) for b in cpu]
    f = BK.BlobFeeder(bk)
    f.submit(pinned[0])
    for t in range(5):
        if t + 1 < 5:
            f.submit(pinned[t + 1])
        f.load_next()
        bk.build_plan_on_device()
        torch.cuda.synchronize()
        assert torch.equal(bk.raw.cpu(), pinned[t]), t
        ok, sizes = bk.check()
        assert ok and sizes["N"] == cpu[t].x.size(0)
