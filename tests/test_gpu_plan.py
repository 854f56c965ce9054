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
from moleculesde_amd import wcache  # noqa: E402

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


@pytest.mark.parametrize("full,per_edge", [(False, False), (True, False), (False, True)], ids=["configs1", "full", "configs1_per_edge_cfconv"])
def test_bucket_step_matches_exact_batch(dev, full, per_edge, monkeypatch):
    """Losses and every parameter gradient of one pretrain step: padded bucket + row bounds + device-built plan vs the
    exact-size batch with the host plan.  Contrastive permutation: identity on both sides is impossible (it is drawn
    on the device), so the contrastive negatives come from a fixed permutation of the VALID atoms.  per_edge: SchNet's CFConv
    on the per-edge kernels (radius CSR + its by-source view built for the PADDED atom count) instead of the pair form."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import bucket as BK, hip, pretrain
    if per_edge:
        monkeypatch.setattr(hip, "CFCONV_PAIR", False)
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
    with bk.bounds():
        parts_b, g_b, sd_b = run(bk.batch, caps.N)
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


def _capturable_noise(G, dev, seed=77, fixed_calls=False):
    """Position noise / time steps from fixed DEVICE tensors (padding independent, capturable); contrastive permutations
    from the device kernel (a function of seed and call count)."""
    class Noise(G.DeviceNoise):
        def __init__(self):
            super().__init__(seed=seed)
            g = torch.Generator().manual_seed(5)
            self.big = torch.randn(8192, 3, generator=g).to(dev)
            self.ints = torch.randint(0, 1000, (4096,), generator=g).to(dev)

        def randn_like(self, x):
            assert x.dim() == 2 and x.size(1) == 3
            return self.big[:x.size(0)].clone()

        def randint(self, high, size, device):
            return self.ints[:size[0]].clone()

        def randperm_pair(self, n, device):
            if fixed_calls:         # the host-side draw counter is baked into a captured graph: keep it out of the seed, so
                self.calls = 0      # that graphs captured at different times draw alike (the device step counter varies them)
            return super().randperm_pair(n, device)
    return Noise()


@pytest.mark.timeout(600)
def test_oversized_batch_falls_back_and_overflow_is_flagged(dev):
    """Hardening of the bucket path: (1) pack_raw refuses a batch that does not fit (host side, free); (2) Trainer.step_stream
    sends such a batch through the exact-size eager step and the result equals a plain Trainer.step on it; (3) a blob
    forced past the host check only sets the device-side flag -- no out-of-bounds write, the flagged rows inert -- and
    step_bucket raises one call later without ever synchronising the running step; (4) Caps rejects molecules the
    device-side plan builder cannot take and keeps clear of other row counts of the step."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import bucket as BK, hip, pretrain
    from moleculesde_amd.synthetic import make_batch
    args = pretrain.readme_args(emb_dim=64, SDE_coeff_generative_3Dto2D=0)
    cand = sorted((make_batch(16, seed=s) for s in range(71, 79)), key=lambda q: BK.raw_sizes(q)["N"])
    small, big = cand[:2], cand[-1]
    top = max(BK.raw_sizes(q)["N"] for q in small)
    # capacities sized for the small batches only (N rounded up to 64 must stay below the big batch)
    need_s = [BK.raw_sizes(q) for q in small]
    caps = BK.Caps.covering(need_s, n_max=24)
    assert BK.raw_sizes(big)["N"] > caps.N or not caps.fits(BK.raw_sizes(big)), (BK.raw_sizes(big), caps.as_dict())
    assert not caps.fits(BK.raw_sizes(big))
    with pytest.raises(BK.BucketOverflow):
        BK.pack_raw(big, caps)
    with pytest.raises(BK.BucketOverflow):
        BK.Caps(16, 100, 100, 100, 100, 100, 40)          # n_max beyond the plan builder
    c2 = BK.Caps(16, 16 * 24 - 10, 100, 100, 100, 100, 24)
    assert c2.N != 16 * 24 and len({c2.N, c2.E_b, c2.E_e, c2.E_r, c2.P}) == 5

    def trainer():
        torch.manual_seed(5)
        t = pretrain.Trainer(args, dev)
        t.overlap_streams = False
        for m in t.models.values():
            disable_dropout(m)
        n = _capturable_noise(G, dev)
        t.noise = n
        t.models["SDE_2Dto3D_model"].noise = n
        if "SDE_3Dto2D_model" in t.models:
            t.models["SDE_3Dto2D_model"].noise = n
        return t
    fits = [b for b in small if caps.fits(BK.raw_sizes(b))]
    assert fits
    tr = trainer()
    bk = tr.make_bucket(caps)
    tr.capture_bucket(bk, BK.pack_raw(fits[0], caps).to(dev))
    snap = (tr.opt.flat_p.clone(), tr.opt.m.clone(), tr.opt.v.clone(), tr.opt.step_dev.clone(), tr.step_counter.clone(),
            {k: {n: v.clone() for n, v in m.state_dict().items()} for k, m in tr.models.items()})

    calls0 = tr.noise.calls

    def restore():
        tr.opt.flat_p.copy_(snap[0]); tr.opt.m.copy_(snap[1]); tr.opt.v.copy_(snap[2]); tr.opt.step_dev.copy_(snap[3])
        tr.step_counter.copy_(snap[4])
        for k, m in tr.models.items():
            m.load_state_dict(snap[5][k])
        tr.noise.calls = calls0
        wcache.bump_weight_epoch()
    tr.step_stream(bk, big)                               # does not fit -> exact-size eager step
    torch.cuda.synchronize()
    got = tr.opt.flat_p.clone()
    restore()
    tr.step(G.prepare_batch(big.clone(), dev))            # the plain exact-size step from the same state
    torch.cuda.synchronize()
    assert torch.isfinite(got).all() and torch.equal(got, tr.opt.flat_p), "fallback == exact-size step"
    tr.step_stream(bk, fits[0])                           # and the bucket keeps working afterwards
    torch.cuda.synchronize()
    ok, _ = bk.check()
    assert ok
    # (3) a blob that lies about its sizes: more atoms than the capacity
    blob = BK.pack_raw(fits[0], caps).clone()
    o, n = bk.layout["mol_atoms"]
    blob[o] = caps.N + 5                                  # first molecule claims more atoms than the bucket holds
    guard = torch.full((4096,), 12345, dtype=torch.int32, device=dev)      # memory next to the bucket's buffers
    tr.step_bucket(bk, blob.to(dev))
    torch.cuda.synchronize()
    assert int(bk.err.cpu()) == 1
    assert bool((guard == 12345).all())
    assert not bk.check()[0]
    # (3b) a blob whose ATOMS fit but whose atom pairs (sum n^2: the rows of the dense head's [P, *] arrays, the CFConv's
    # pair list) do not: few, large molecules.  The plan cuts the batch in front of the first molecule that does not fit,
    # every derived size stays inside its capacity, and the flag is raised
    blob2 = BK.pack_raw(fits[0], caps).clone()
    k = min(caps.B, caps.N // 24)
    assert k * 24 * 24 > caps.P, (k, caps.as_dict())
    blob2[o:o + caps.B] = torch.tensor([24] * k + [0] * (caps.B - k), dtype=torch.int32)
    try:
        tr.step_bucket(bk, blob2.to(dev))
    except BK.BucketOverflow:
        tr.step_bucket(bk, blob2.to(dev))                     # (the poll of the previous bad step)
    torch.cuda.synchronize()
    sz = bk.sizes.cpu()
    assert int(bk.err.cpu()) == 1
    assert int(sz[0]) <= caps.N and int(sz[3]) <= caps.P and int(sz[5]) <= caps.E_r, (sz.tolist(), caps.as_dict())
    assert int(bk.pair_ptr.cpu().max()) <= caps.P
    assert bool((guard == 12345).all())
    try:
        tr.step_bucket(bk, BK.pack_raw(fits[0], caps).to(dev))   # poll is one call late: this call only queues the copy ...
    except BK.BucketOverflow:
        pass
    with pytest.raises(BK.BucketOverflow):
        for _ in range(3):                                    # ... and one of the next calls reports the bad step
            tr.step_bucket(bk, BK.pack_raw(fits[0], caps).to(dev))
    hip.clear_row_bounds()


@pytest.mark.timeout(600)
def test_two_live_buckets(dev):
    """Two buckets of different capacities, each with its own captured graph, used alternately: the row bounds a graph was
    captured with stay its own (device pointers baked in at capture), so interleaving them gives the same parameters as
    using each alone."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import bucket as BK, hip, pretrain
    from moleculesde_amd.synthetic import make_batch
    args = pretrain.readme_args(emb_dim=64, SDE_coeff_generative_3Dto2D=0)
    A = [make_batch(12, seed=s) for s in (81, 82)]
    Bb = [make_batch(24, seed=s) for s in (83, 84)]

    def run(order):
        torch.manual_seed(9)
        tr = pretrain.Trainer(args, dev)
        tr.overlap_streams = False
        for m in tr.models.values():
            disable_dropout(m)
        n = _capturable_noise(G, dev)
        tr.noise = n
        tr.models["SDE_2Dto3D_model"].noise = n
        ca, cb = BK.Caps.covering([BK.raw_sizes(b) for b in A], n_max=24), BK.Caps.covering([BK.raw_sizes(b) for b in Bb], n_max=24)
        ba, bb = tr.make_bucket(ca), tr.make_bucket(cb)
        blobs = {"a": [BK.pack_raw(b, ca).to(dev) for b in A], "b": [BK.pack_raw(b, cb).to(dev) for b in Bb]}
        p0 = tr.opt.flat_p.clone()
        tr.capture_bucket(ba, blobs["a"][0])
        tr.capture_bucket(bb, blobs["b"][0])
        # capture warm-up stepped the optimiser: back to a common starting point
        tr.opt.flat_p.copy_(p0); tr.opt.m.zero_(); tr.opt.v.zero_(); tr.opt.step_dev.zero_(); tr.step_counter.zero_()
        wcache.refresh_weight_t()
        for which, i in order:
            tr.step_bucket(ba if which == "a" else bb, blobs[which][i])
        torch.cuda.synchronize()
        assert ba.check()[0] and bb.check()[0]
        hip.clear_row_bounds()
        return tr.opt.flat_p.clone()
    inter = run([("a", 0), ("b", 0), ("a", 1), ("b", 1)])
    again = run([("a", 0), ("b", 0), ("a", 1), ("b", 1)])
    assert torch.equal(inter, again), "same schedule, same parameters (bitwise)"
    assert torch.isfinite(inter).all()


@pytest.mark.timeout(600)
def test_bucket_pipeline_matches_single_bucket(dev):
    """pretrain.BucketPipeline (two buckets used alternately, the plans of batch t+1 built by a captured plan graph on a
    third stream while the step of batch t runs) gives bit for bit the parameters of the one-bucket loop over the same
    blobs: same capacities => same kernels, same split geometry, same order."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import bucket as BK, hip, pretrain
    from moleculesde_amd.synthetic import make_batch
    args = pretrain.readme_args(emb_dim=64, SDE_coeff_generative_3Dto2D=0)
    cpu = [make_batch(16, seed=s) for s in (91, 92, 93, 94, 95)]
    caps = BK.Caps.covering([BK.raw_sizes(b) for b in cpu])

    def run(piped):
        torch.manual_seed(9)
        tr = pretrain.Trainer(args, dev)
        for m in tr.models.values():
            disable_dropout(m)
        n = _capturable_noise(G, dev, fixed_calls=True)
        tr.noise = n
        tr.models["SDE_2Dto3D_model"].noise = n
        blobs = [BK.pack_raw(b, caps).to(dev) for b in cpu]
        p0 = tr.opt.flat_p.clone()
        if piped:
            pipe = pretrain.BucketPipeline(tr, caps, blobs[0])
        else:
            bk = tr.make_bucket(caps)
            tr.capture_bucket(bk, blobs[0])
        tr.opt.flat_p.copy_(p0); tr.opt.m.zero_(); tr.opt.v.zero_(); tr.opt.step_dev.zero_(); tr.step_counter.zero_()
        wcache.refresh_weight_t()
        losses = []
        if piped:
            pipe.submit(blobs[0])
            for t in range(len(blobs)):
                if t + 1 < len(blobs):
                    pipe.submit(blobs[t + 1])
                losses.append(pipe.step().clone())
            torch.cuda.synchronize()
            assert pipe.check()
        else:
            for b in blobs:
                losses.append(tr.step_bucket(bk, b).clone())
            torch.cuda.synchronize()
            assert bk.check()[0]
        hip.clear_row_bounds()
        return tr.opt.flat_p.clone(), torch.stack(losses)
    p_one, l_one = run(False)
    p_two, l_two = run(True)
    assert torch.isfinite(p_one).all() and torch.isfinite(l_one).all()
    assert torch.equal(l_one, l_two), (l_one, l_two)
    assert torch.equal(p_one, p_two), float((p_one - p_two).abs().max())


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


@pytest.mark.timeout(900)
def test_stream_runner_matches_its_eager_twin_and_main_runs(dev, capsys):
    """the product's own training loop (pretrain.main / train_epochs / StreamRunner -- the counterpart
    of examples/pretrain_MoleculeSDE.py:106-175) drives the captured-graph + capacity-bucket + prefetch path.  (1) the
    replayed loop and its eager twin (same bucket, same device-side plans, kernels launched from the host) give the same
    per-epoch losses and the same parameters; (2) a contrastive skip epoch re-captures with the new coefficient; (3) main()
    runs end to end and prints the reference's epoch report."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import pretrain
    from moleculesde_amd.synthetic import make_batch
    args = pretrain.readme_args(emb_dim=64, SDE_coeff_generative_3Dto2D=0, batch_size=12, epochs=2)
    args.SDE_coeff_contrastive_skip_epochs = 1
    args.output_model_dir = ""
    cpu = [make_batch(12, seed=s) for s in range(201, 206)]
    res = []
    for replay in (True, False):
        torch.manual_seed(11)
        tr = pretrain.Trainer(args, dev)
        for m in tr.models.values():
            disable_dropout(m)
        nz = _capturable_noise(G, dev, fixed_calls=True)          # the same draws whether a step is replayed or launched
        tr.noise = nz
        tr.models["SDE_2Dto3D_model"].noise = nz
        runner = pretrain.StreamRunner(tr, cpu, n_max=24, replay=replay)
        blobs = [runner.pack(b) for b in cpu]
        assert all(b is not None for b in blobs)
        hist = pretrain.train_epochs(args, tr, runner, lambda e: (blobs, 7), out=lambda *_: None)
        torch.cuda.synchronize()
        res.append((hist, tr.opt.flat_p.clone()))
        from moleculesde_amd import hip
        hip.clear_row_bounds()
    (ha, pa), (hb, pb) = res
    assert ha[0][0] == 0.0 and ha[1][0] > 0.0, "contrastive loss skipped in epoch 1, on in epoch 2"
    for ea, eb in zip(ha, hb):
        for x, y in zip(ea[:4], eb[:4]):
            assert abs(x - y) <= 2e-5 * max(1.0, abs(y)), (ea, eb)
    assert float((pa - pb).norm() / pb.norm()) < 1e-5
    # (3) the command-line entry point, tiny configuration
    out = pretrain.main(["--epochs", "1", "--steps_per_epoch", "5", "--batch_size", "8", "--emb_dim", "64", "--synthetic_pool", "3",
                         "--SDE_2Dto3D_model", "SDEModel2Dto3D_02", "--CL_similarity_metric", "EBM_node_dot_prod",
                         "--SDE_coeff_generative_3Dto2D", "0", "--dropout_ratio", "0"])
    printed = capsys.readouterr().out
    assert "epoch: 1" in printed and "SDE 2Dto3D Loss" in printed and len(out) == 1 and out[0][5] == 5
    import math
    assert all(math.isfinite(v) for v in out[0][:4])


def test_stream_runner_takes_the_exact_step_for_a_batch_outside_the_bucket(dev):
    """a batch that does not fit the capacity bucket (here: more molecules than the bucket was built for) must
    take the documented fallback -- the exact-size eager step on its own host-built plan -- not crash in pack(None)."""
    from moleculesde_amd import hip, pretrain
    from moleculesde_amd.synthetic import make_batch
    args = pretrain.readme_args(emb_dim=64, SDE_coeff_generative_3Dto2D=0, batch_size=12)
    torch.manual_seed(3)
    tr = pretrain.Trainer(args, dev)
    cpu = [make_batch(12, seed=s) for s in range(301, 304)]
    big = make_batch(20, seed=399)
    runner = pretrain.StreamRunner(tr, cpu, n_max=24)
    items = [runner.prepare(b) for b in cpu[:2]] + [runner.prepare(big), runner.prepare(cpu[2])]
    assert items[2][1] is None and all(it[1] is not None for it in (items[0], items[1], items[3]))
    p0 = tr.opt.flat_p.clone()
    assert runner.run(items) == 4 and runner.fallbacks == 1
    assert runner.run([cpu[0], big]) == 2 and runner.fallbacks == 2          # bare host batches are packed on the fly
    with pytest.raises(ValueError):
        runner.run([None])
    torch.cuda.synchronize()
    assert torch.isfinite(tr.opt.flat_p).all() and not torch.equal(tr.opt.flat_p, p0)
    hip.clear_row_bounds()
