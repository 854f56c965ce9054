"""Kernel-level parity (GPU box): every C-ABI entry point, called through moleculesde_amd.hip, against
the oracle's CPU restatement of the same op on seeded inputs.  Integer/index outputs are bit-exact;
floating point is fp32 with the tolerance written at each assert."""
import math

import pytest
import torch
from moleculesde_amd import slabs, wcache  # noqa: E402

pytestmark = pytest.mark.gpu

from oracle import restate as R  # noqa: E402
from helpers import assert_close  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from moleculesde_amd import _lib
    _lib.load()  # fails loudly if libmsde_hip.so is missing
    return torch.device("cuda", 0)


def _toy_graph(seed=0, B=6):
    from moleculesde_amd.synthetic import make_batch
    return make_batch(B, seed)


def _plan_for(edge_index, N, dev):
    from moleculesde_amd import hip
    return hip.build_csr(edge_index, N).to(dev)


# ------------------------------------------------------------------------------------------------
def test_abi_identity(dev):
    from moleculesde_amd import _lib
    lib = _lib.load()
    assert lib.msde_abi_version() == 1
    assert lib.msde_target_arch() == b"gfx950"


@pytest.mark.parametrize("n", [0, 1, 63, 64, 1023, 1024, 1025, 5000])
def test_exclusive_scan(dev, n):
    from moleculesde_amd import _lib, hip
    g = torch.Generator().manual_seed(n)
    x = torch.randint(0, 33, (n,), generator=g, dtype=torch.int32)
    xd = x.to(dev)
    out = torch.empty(n + 1, dtype=torch.int32, device=dev)
    _lib.call("msde_exclusive_scan_i32", hip._p(xd), hip._p(out), n, hip._stream())
    ref = torch.cat([torch.zeros(1, dtype=torch.int64), x.long().cumsum(0)])
    assert torch.equal(out.cpu().long(), ref)


def _radius_check(dev, pos, batch, cutoff, max_nbr=32):
    from moleculesde_amd import hip, plan as P
    import types
    d = types.SimpleNamespace(x=torch.zeros(pos.size(0), dtype=torch.long), batch=batch, num_graphs=int(batch.max()) + 1,
                              edge_index=None, edge_attr=None)
    pl = P.plan_to(P.build_plan(d, max_nbr=max_nbr, with_ext=False), dev)
    rp, dist = hip.radius_plan(pos.to(dev), pl.batch_i32, pl.mol_ptr, cutoff, pl.E_r_cap, max_nbr)
    ei = R.radius_graph(pos, cutoff, batch, max_nbr)
    E = ei.size(1)
    assert int(rp.rowptr[-1]) == E
    assert torch.equal(rp.src[:E].cpu().long(), ei[0])          # grouped by target, index order
    assert torch.equal(rp.dst[:E].cpu().long(), ei[1])
    assert bool((rp.src[E:] == -1).all()) and bool((dist[E:] == 0).all())
    dref = (pos[ei[0]] - pos[ei[1]]).norm(dim=-1)
    assert_close(dist[:E], dref, 1e-6, 1e-7, "edge distance")
    # transposed view is a permutation of the real edges, grouped by source
    perm = rp.perm_s[:E].cpu().long()
    assert torch.equal(torch.sort(perm)[0], torch.arange(E))
    s_sorted = ei[0][perm]
    assert torch.equal(s_sorted, torch.sort(ei[0], stable=True)[0])
    N = pos.size(0)
    assert torch.equal(rp.rowptr_s.cpu().long(), torch.searchsorted(s_sorted, torch.arange(N + 1)))
    if 0 < pl.N_max <= hip.RADIUS_TRANSPOSE_MOL_NMAX:
        # the one-launch form (one workgroup per molecule): identical tables, padded slots included
        rp2, _ = hip.radius_plan(pos.to(dev), pl.batch_i32, pl.mol_ptr, cutoff, pl.E_r_cap, max_nbr, n_max=pl.N_max)
        assert torch.equal(rp2.rowptr_s, rp.rowptr_s) and torch.equal(rp2.perm_s, rp.perm_s)
    return rp, dist


def test_radius_graph_synthetic(dev):
    b = _toy_graph(3, 16)
    _radius_check(dev, b.positions, b.batch, 10.0)
    _radius_check(dev, b.positions, b.batch, 2.0)      # sparse: most pairs outside


def test_radius_graph_strict_and_cap(dev):
    pos = torch.tensor([[0.0, 0, 0], [1.0, 0, 0], [0.0, 0, 0], [11.0, 0, 0], [10.0, 0, 0]])
    batch = torch.tensor([0, 0, 1, 1, 1])
    _radius_check(dev, pos, batch, 10.0)
    line = torch.zeros(40, 3)
    line[:, 0] = torch.arange(40) * 0.01
    rp, _ = _radius_check(dev, line, torch.zeros(40, dtype=torch.long), 10.0, 32)   # cap binds: asymmetric graph
    assert int((rp.rowptr[1:] - rp.rowptr[:-1]).max()) == 32


@pytest.mark.parametrize("D", [300, 128, 32, 3])
def test_segment_sum_and_gathers(dev, D):
    from moleculesde_amd import hip
    torch.manual_seed(D)
    b = _toy_graph(1, 8)
    N = b.x.size(0)
    pl = _plan_for(b.extended_edge_index, N, dev)
    E = pl.E
    rows = torch.randn(E, D)
    rd = rows.to(dev)
    dst = pl.dst.cpu().long()
    src = pl.src.cpu().long()
    out_t = hip.segment_sum_rows(rd, pl.rowptr, None, N)
    assert_close(out_t, R.scatter_sum(rows, dst, N), 1e-5, 1e-5, "segment sum by target")
    out_s = hip.segment_sum_rows(rd, pl.rowptr_s, pl.perm_s, N)
    assert_close(out_s, R.scatter_sum(rows, src, N), 1e-5, 1e-5, "segment sum by source")
    out_m = hip.segment_sum_rows(rd, pl.rowptr, None, N, mean=True)
    assert_close(out_m, R.scatter_mean(rows, dst, N), 1e-5, 1e-5, "segment mean")
    A, Bm = torch.randn(N, D), torch.randn(N, D)
    pg = hip.pair_gather_add(A.to(dev), Bm.to(dev), pl)
    assert_close(pg, A[src] + Bm[dst], 0, 0, "pair gather add")     # exact: one add
    assert_close(hip.gather_rows(A.to(dev), pl.src), A[src], 0, 0, "gather rows")


def test_pair_gather_add_backward(dev):
    from moleculesde_amd import hip
    torch.manual_seed(5)
    b = _toy_graph(2, 8)
    N = b.x.size(0)
    pl = _plan_for(b.extended_edge_index, N, dev)
    src, dst = pl.src.cpu().long(), pl.dst.cpu().long()
    A = torch.randn(N, 32, requires_grad=True)
    Bm = torch.randn(N, 32, requires_grad=True)
    w = torch.randn(pl.E, 32)
    ((A[src] + Bm[dst]) * w).sum().backward()
    Ad, Bd = A.detach().to(dev).requires_grad_(True), Bm.detach().to(dev).requires_grad_(True)
    (hip.pair_gather_add(Ad, Bd, pl) * w.to(dev)).sum().backward()
    assert_close(Ad.grad, A.grad, 1e-5, 1e-5, "gA")
    assert_close(Bd.grad, Bm.grad, 1e-5, 1e-5, "gB")


@pytest.mark.parametrize("B,D,H", [(8, 300, 32), (40, 300, 32), (40, 64, 64), (3, 32, 32)])
def test_pair_bn_relu_linear_fused(dev, B, D, H):
    """hip._PairBnReluLinear (edge_2D_emb of the 2D->3D model: gather-add -> BatchNorm1d(train) -> ReLU -> Linear around ONE
    gather, SDE_model_2D_to_3D.py:35-40,264-271) against fp64 autograd: output, running statistics, and the gradients of AB,
    the BatchNorm affine parameters and the second Linear; few edges (row-strip products) and > 512 edges (2-D tiles)."""
    from moleculesde_amd import hip
    from moleculesde_amd.geom3d import nn as mnn
    torch.manual_seed(B * 7 + D)
    b = _toy_graph(3, B)
    N = b.x.size(0)
    pl = _plan_for(b.extended_edge_index, N, dev)
    src, dst = pl.src.cpu().long(), pl.dst.cpu().long()
    E = pl.E
    AB = (torch.randn(N, 2 * D) * 1.5 + 0.3).double().requires_grad_(True)
    bn = torch.nn.BatchNorm1d(D).double()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
    lin = torch.nn.Linear(D, H).double()
    R = torch.randn(E, H).double()
    out_ref = lin(torch.relu(bn(AB[src, :D] + AB[dst, D:])))
    (out_ref * R).sum().backward()
    bnd = mnn.BatchNorm1d(D).to(dev)
    bnd.fuse_relu = True
    lind = mnn.Linear(D, H).to(dev)
    with torch.no_grad():
        bnd.weight.copy_(bn.weight.float()); bnd.bias.copy_(bn.bias.float())
        lind.weight.copy_(lin.weight.float()); lind.bias.copy_(lin.bias.float())
    ABd = AB.detach().float().to(dev).requires_grad_(True)
    assert mnn.bn_fusable(bnd) and hip.pair_bn_relu_linear_ok(ABd, bnd, lind)
    out = hip.pair_bn_relu_linear(ABd, pl, bnd, lind)
    (out * R.float().to(dev)).sum().backward()
    assert_close(out, out_ref, 2e-4, 2e-4, "out")
    assert_close(bnd.running_mean, bn.running_mean, 1e-5, 1e-5, "running_mean")
    assert_close(bnd.running_var, bn.running_var, 1e-4, 1e-5, "running_var")
    sc = float(AB.grad.abs().max())
    assert_close(ABd.grad, AB.grad, 2e-3, 2e-4 * sc, "g_AB")
    assert_close(bnd.weight.grad, bn.weight.grad, 2e-3, 2e-4 * float(bn.weight.grad.abs().max()), "dgamma")
    assert_close(bnd.bias.grad, bn.bias.grad, 2e-3, 2e-4 * float(bn.bias.grad.abs().max()), "dbeta")
    assert_close(lind.weight.grad, lin.weight.grad, 2e-3, 2e-4 * float(lin.weight.grad.abs().max()), "gW2")
    assert_close(lind.bias.grad, lin.bias.grad, 2e-3, 2e-4 * float(lin.bias.grad.abs().max()), "gb2")
    # and the unfused operator chain of the same modules gives the same numbers
    AB2 = AB.detach().float().to(dev).requires_grad_(True)
    bn2 = mnn.BatchNorm1d(D).to(dev); bn2.fuse_relu = True
    with torch.no_grad():
        bn2.weight.copy_(bn.weight.float()); bn2.bias.copy_(bn.bias.float())
    out2 = lind(bn2(hip.pair_gather_add_cols(AB2, pl)))
    assert_close(out, out2, 1e-4, 1e-4, "fused vs operator chain")


@pytest.mark.parametrize("B,with_base", [(6, True), (40, False)])
def test_mlp_head_mix_fused(dev, B, with_base):
    """hip._MlpHeadMix (basis MLP + frame mix + mean over in-edges + running sum, equivariant_scorenetwork.py:142-166, one
    autograd node) against fp64 autograd and against the two-operator form (hip.mlp_fused + hip.frame_mix_mean): output and the
    gradients of the input, of both Linear layers and of the running sum -- inside a parameter-gradient batch (queued slabs)
    and outside one."""
    from moleculesde_amd import hip
    from moleculesde_amd.geom3d import nn as mnn
    torch.manual_seed(B)
    b = _toy_graph(4, B)
    N = b.x.size(0)
    pl = _plan_for(b.extended_edge_index, N, dev)
    E = pl.E
    dst = pl.dst.cpu().long()
    x = torch.randn(E, 64).double().requires_grad_(True)
    basis = torch.randn(E, 3, 3).double()
    base = torch.randn(N, 3).double().requires_grad_(True) if with_base else None
    l1, l2 = torch.nn.Linear(64, 128).double(), torch.nn.Linear(128, 3).double()
    coff = l2(torch.nn.functional.silu(l1(x)))
    mixed = (coff[:, :, None] * basis).sum(1)
    deg = torch.zeros(N, dtype=torch.float64).index_add_(0, dst, torch.ones(E, dtype=torch.float64)).clamp(min=1)
    ref = torch.zeros(N, 3, dtype=torch.float64).index_add_(0, dst, mixed) / deg[:, None]
    if base is not None:
        ref = ref + base
    R = torch.randn(N, 3).double()
    (ref * R).sum().backward()
    for batched in (False, True):
        d1, d2 = mnn.Linear(64, 128).to(dev), mnn.Linear(128, 3).to(dev)
        with torch.no_grad():
            d1.weight.copy_(l1.weight.float()); d1.bias.copy_(l1.bias.float())
            d2.weight.copy_(l2.weight.float()); d2.bias.copy_(l2.bias.float())
        xd = x.detach().float().to(dev).requires_grad_(True)
        based = base.detach().float().to(dev).requires_grad_(True) if base is not None else None
        bd = basis.float().to(dev).contiguous()
        assert hip.mlp_head_mix_ok(xd, d1, d2)
        if batched:
            slabs.begin_param_grad_batch([d1.weight, d1.bias, d2.weight, d2.bias])
        try:
            out = hip.mlp_head_mix(xd, d1, d2, bd.view(E, 9), pl, based)
            (out * R.float().to(dev)).sum().backward()
        finally:
            if batched:
                slabs.flush_wgrad_gemms()
                slabs.finish_param_grad_batch()
        assert_close(out, ref, 1e-4, 1e-4, "out")
        assert_close(xd.grad, x.grad, 2e-3, 1e-5, "g_x")
        assert_close(d1.weight.grad, l1.weight.grad, 2e-3, 2e-4 * float(l1.weight.grad.abs().max()), "gW1")
        assert_close(d1.bias.grad, l1.bias.grad, 2e-3, 2e-4 * float(l1.bias.grad.abs().max()), "gb1")
        assert_close(d2.weight.grad, l2.weight.grad, 2e-3, 2e-4 * float(l2.weight.grad.abs().max()), "gW2")
        assert_close(d2.bias.grad, l2.bias.grad, 2e-3, 2e-4 * float(l2.bias.grad.abs().max()), "gb2")
        if based is not None:
            assert_close(based.grad, base.grad, 1e-6, 1e-6, "g_base")
        # the two-operator form gives the same output (same order of additions; multiply-add contraction may differ)
        with torch.no_grad():
            c2 = hip.mlp_fused(xd.detach(), [(d1.weight, d1.bias), (d2.weight, d2.bias)], "silu")
            o2 = hip.frame_mix_mean(c2, bd.view(E, 9), pl, based.detach() if based is not None else None)
        assert_close(out.detach(), o2, 1e-6, 1e-6, "fused vs two-operator output")


@pytest.mark.parametrize("J", [3, 32])
def test_pair_gather_cat_and_fused_mlp(dev, J):
    """cat([h_row + h_col, edge_attr]) -> Linear -> SiLU -> Linear (equivariant_scorenetwork.py:154-157, 142-146): the
    gather-written concatenation + the gemm_ex MLP against the plain fp32 torch operators (forward and every gradient,
    immediate and batched weight-gradient paths)."""
    from moleculesde_amd import hip
    torch.manual_seed(11)
    b = _toy_graph(6, 12)
    N = b.x.size(0)
    pl = _plan_for(b.extended_edge_index, N, dev)
    src, dst = pl.src.cpu().long(), pl.dst.cpu().long()
    D, H = 32, 128
    h = torch.randn(N, D, requires_grad=True)
    ea = torch.randn(pl.E, D, requires_grad=True)
    l0, l1 = torch.nn.Linear(2 * D, H), torch.nn.Linear(H, J)     # J = 3: the row-kernel head; 32: GEMM epilogues only
    w = torch.randn(pl.E, J)
    ref = l1(torch.nn.functional.silu(l0(torch.cat([h[src] + h[dst], ea], -1))))
    (ref * w).sum().backward()
    for batched in (False, True):
        hd, ed = h.detach().to(dev).requires_grad_(True), ea.detach().to(dev).requires_grad_(True)
        ps = [p.detach().to(dev).requires_grad_(True) for p in (l0.weight, l0.bias, l1.weight, l1.bias)]
        if batched:
            slabs.begin_param_grad_batch(ps)
        out = hip.mlp_fused(hip.pair_gather_cat(hd, ed, pl), [(ps[0], ps[1]), (ps[2], ps[3])], "silu")
        assert_close(out, ref, 2e-5, 2e-5, "fused basis MLP fwd")
        (out * w.to(dev)).sum().backward()
        if batched:
            slabs.finish_param_grad_batch()
        assert_close(hd.grad, h.grad, 1e-4, 1e-4, "g h")
        assert_close(ed.grad, ea.grad, 1e-4, 1e-4, "g edge_attr")
        for got, want, name in zip(ps, (l0.weight, l0.bias, l1.weight, l1.bias), ("W0", "b0", "W1", "b1")):
            assert_close(got.grad, want.grad, 2e-4, 2e-4, "g " + name)


@pytest.mark.parametrize("D", [300, 16])
def test_embedding_sum(dev, D):
    from moleculesde_amd import hip, plan as P
    torch.manual_seed(1)
    b = _toy_graph(4, 32)
    pl = P.plan_to(P.build_plan(b), dev)
    enc = R._EmbeddingSum(R.ATOM_FEATURE_DIMS, D, "atom_embedding_list")
    ref = enc(b.x)
    w = torch.randn_like(ref)
    (ref * w).sum().backward()
    tab = torch.cat([e.weight.detach() for e in enc.atom_embedding_list]).to(dev).requires_grad_(True)
    out = hip.embedding_sum(tab, pl.atom_codes, pl.atom_list_ptr, pl.atom_list_nodes)
    assert_close(out, ref, 1e-6, 1e-6, "atom encoder fwd")
    (out * w.to(dev)).sum().backward()
    gref = torch.cat([e.weight.grad for e in enc.atom_embedding_list])
    assert_close(tab.grad, gref, 1e-4, 1e-4, "atom encoder table grad")


@pytest.mark.parametrize("D", [300, 16, 6])
def test_gin_aggregate(dev, D):
    from moleculesde_amd import hip, plan as P
    torch.manual_seed(2)
    b = _toy_graph(5, 32)
    pl = P.plan_to(P.build_plan(b), dev)
    conv = R.GINConv(D)
    conv.mlp = torch.nn.Identity()
    with torch.no_grad():
        conv.eps.fill_(0.3)
    x = torch.randn(b.x.size(0), D, requires_grad=True)
    ref = conv(x, b.edge_index, b.edge_attr)
    w = torch.randn_like(ref)
    (ref * w).sum().backward()
    tab = torch.cat([e.weight.detach() for e in conv.bond_encoder.bond_embedding_list]).to(dev).requires_grad_(True)
    xd = x.detach().to(dev).requires_grad_(True)
    eps = conv.eps.detach().to(dev).requires_grad_(True)
    out = hip.gin_aggregate(xd, tab, eps, pl.bond, pl.bond_codes)
    assert_close(out, ref, 1e-5, 1e-5, "gin fwd")
    (out * w.to(dev)).sum().backward()
    assert_close(xd.grad, x.grad, 1e-5, 1e-5, "gin g_x")
    gtab = torch.cat([e.weight.grad for e in conv.bond_encoder.bond_embedding_list])
    assert_close(tab.grad, gtab, 1e-4, 1e-4, "gin g_tab")
    assert_close(eps.grad, conv.eps.grad, 1e-4, 1e-4, "gin g_eps")


def _radius_setup(dev, B=24, seed=7):
    from moleculesde_amd import hip, plan as P
    b = _toy_graph(seed, B)
    pl = P.plan_to(P.build_plan(b), dev)
    rp, dist = hip.radius_plan(b.positions.to(dev), pl.batch_i32, pl.mol_ptr, 10.0, pl.E_r_cap, 32)
    E = int(rp.rowptr[-1])
    ei = torch.stack([rp.src[:E].cpu().long(), rp.dst[:E].cpu().long()])
    return b, pl, rp, dist, E, ei


def test_rbf_cutoff(dev):
    from moleculesde_amd import hip
    b, pl, rp, dist, E, ei = _radius_setup(dev)
    gs = R.GaussianSmearing(0.0, 10.0, 51)
    rbf, C = hip.rbf_cutoff(dist, rp.E_dev, gs.offset.to(dev), gs.coeff, 10.0)
    d = dist[:E].cpu()
    assert_close(rbf[:E], gs(d), 1e-5, 1e-6, "rbf")
    assert_close(C[:E], 0.5 * (torch.cos(d * math.pi / 10.0) + 1.0), 1e-5, 1e-6, "cutoff")
    assert bool((rbf[E:] == 0).all()) and bool((C[E:] == 0).all())


@pytest.mark.parametrize("Fd", [128, 16])
def test_cfconv_aggregate(dev, Fd):
    from moleculesde_amd import hip
    torch.manual_seed(3)
    b, pl, rp, dist, E, ei = _radius_setup(dev)
    N = b.x.size(0)
    x1 = torch.randn(N, Fd, requires_grad=True)
    Wf = torch.randn(rp.E, Fd, requires_grad=True)
    C = torch.rand(rp.E)
    ref = R.scatter_sum(x1[ei[0]] * (Wf[:E] * C[:E, None]), ei[1], N)
    w = torch.randn_like(ref)
    (ref * w).sum().backward()
    x1d = x1.detach().to(dev).requires_grad_(True)
    Wfd = Wf.detach().to(dev).requires_grad_(True)
    out = hip.cfconv_aggregate(x1d, Wfd, C.to(dev), rp)
    assert_close(out, ref, 1e-5, 1e-5, "cfconv fwd")
    (out * w.to(dev)).sum().backward()
    assert_close(x1d.grad, x1.grad, 1e-5, 1e-5, "cfconv g_x1")
    assert_close(Wfd.grad, Wf.grad, 1e-5, 1e-5, "cfconv g_Wf (padded rows zero)")


@pytest.mark.parametrize("G,cpw", [(51, 1), (50, 3), (20, 2)])
def test_cfconv_fused_forward_and_backward(dev, G, cpw):
    """Fused fp32-MFMA CFConv (forward, filter rows, g_x1, and the four filter-network weight gradients
    recomputed on chip) vs autograd through the oracle's CFConv interior (schnet.py:141-145,185-195)."""
    from moleculesde_amd import hip
    torch.manual_seed(4)
    b, pl, rp, dist, E, ei = _radius_setup(dev, B=40, seed=9)
    N = b.x.size(0)
    blk = R.InteractionBlock(hidden_channels=64, num_gaussians=G, num_filters=128, cutoff=10.0)
    with torch.no_grad():
        blk.mlp[0].bias.normal_(0, 0.1)
        blk.mlp[2].bias.normal_(0, 0.1)
    gs = R.GaussianSmearing(0.0, 10.0, G)
    x1 = torch.randn(N, 128, requires_grad=True)
    d = dist[:E].cpu()
    Cc = 0.5 * (torch.cos(d * math.pi / 10.0) + 1.0)
    Wf_ref = blk.mlp(gs(d)) * Cc.view(-1, 1)
    ref = R.scatter_sum(x1[ei[0]] * Wf_ref, ei[1], N)
    w = torch.randn_like(ref)
    (ref * w).sum().backward()
    ps = [blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight, blk.mlp[2].bias]
    pd = [p.detach().to(dev).requires_grad_(True) for p in ps]
    x1d = x1.detach().to(dev).requires_grad_(True)
    old = hip.FUSED_CHUNKS_PER_WG
    hip.FUSED_CHUNKS_PER_WG = cpw
    try:
        out = hip.cfconv_fused(x1d, pd[0], pd[1], pd[2], pd[3], dist, rp, gs.offset.to(dev), gs.coeff, 10.0)
        out2, Wf = hip.cfconv_fused_forward(x1d.detach(), dist, rp, pd[0].detach(), pd[1].detach(), pd[2].detach(),
                                            pd[3].detach(), gs.offset.to(dev), gs.coeff, 10.0, want_filter=True)
    finally:
        hip.FUSED_CHUNKS_PER_WG = old
    scale = float(ref.abs().max())
    assert_close(out, ref.detach(), 1e-4, 1e-4 * scale, "fused cfconv fwd")
    assert torch.equal(out, out2)                                     # bitwise reproducible (2-addend atomics)
    assert_close(Wf[:E], Wf_ref.detach(), 1e-4, 1e-5, "filter rows")
    (out * w.to(dev)).sum().backward()
    assert_close(x1d.grad, x1.grad, 1e-4, 1e-4 * float(x1.grad.abs().max()), "fused g_x1")
    for name, a, r in zip(("gW1", "gb1", "gW2", "gb2"), pd, ps):
        assert_close(a.grad, r.grad, 1e-3, 2e-4 * float(r.grad.abs().max()), f"fused {name}")


@pytest.mark.parametrize("G,cutoff,stretch,pad", [(51, 10.0, 1.0, 0), (50, 10.0, 1.0, 0), (20, 4.0, 1.0, 0), (51, 10.0, 2.5, 0),
                                                  (51, 10.0, 1.0, 37)])
def test_cfconv_pair_forward_and_backward(dev, G, cutoff, stretch, pad):
    """CFConv on unordered pairs (csrc/cfconv_pair.hip: pair list, filter rows on the matrix cores, fixed-order
    aggregation; backward = the same aggregation of the gradient + the pair form of the recomputing weight-gradient
    kernel) vs autograd through the oracle's CFConv interior on the oracle's radius graph (schnet.py:91-93,141-145,
    185-195).  A short cutoff / stretched coordinates put many pairs BEYOND the cutoff (their rows must contribute
    nothing); `pad` appends atoms of the empty padding molecule as a capacity bucket does."""
    from moleculesde_amd import hip, plan as P
    torch.manual_seed(4)
    b = _toy_graph(9, 40)
    b.positions = b.positions * stretch
    pl = P.plan_to(P.build_plan(b), dev)
    N = b.x.size(0)
    ei = R.radius_graph(b.positions, cutoff, b.batch)                    # oracle stand-in: [source; target], strict <
    dvec = (b.positions[ei[0]] - b.positions[ei[1]]).norm(dim=-1)
    full = int((torch.bincount(b.batch) * (torch.bincount(b.batch) - 1)).sum())
    if cutoff < 10.0 or stretch > 1.0:
        assert ei.size(1) < full, "this case is meant to have pairs beyond the cutoff"
    blk = R.InteractionBlock(hidden_channels=64, num_gaussians=G, num_filters=128, cutoff=cutoff)
    with torch.no_grad():
        blk.mlp[0].bias.normal_(0, 0.1)
        blk.mlp[2].bias.normal_(0, 0.1)
    gs = R.GaussianSmearing(0.0, cutoff, G)
    x1 = torch.randn(N, 128, requires_grad=True)
    Cc = 0.5 * (torch.cos(dvec * math.pi / cutoff) + 1.0)
    Wf_ref = blk.mlp(gs(dvec)) * Cc.view(-1, 1)
    ref = R.scatter_sum(x1[ei[0]] * Wf_ref, ei[1], N)
    w = torch.randn_like(ref)
    (ref * w).sum().backward()
    ps = [blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight, blk.mlp[2].bias]
    pd = [q.detach().to(dev).requires_grad_(True) for q in ps]
    pos_d = b.positions.to(dev)
    x1_full = x1.detach()
    w_full = w
    if pad:      # capacity padding: extra atoms owned by molecule B (no pairs), larger pair capacity
        pl.batch_i32 = torch.cat([pl.batch_i32, torch.full((pad,), pl.B, dtype=torch.int32, device=dev)])
        pl.mol_ptr = torch.cat([pl.mol_ptr, torch.tensor([N + pad], dtype=torch.int32, device=dev)])[:pl.B + 1]
        pos_d = torch.cat([pos_d, torch.randn(pad, 3, device=dev)])
        x1_full = torch.cat([x1_full, torch.randn(pad, 128)])
        w_full = torch.cat([w, torch.randn(pad, 128)])
        pl.P2_cap = pl.P2_cap + 100
    x1d = x1_full.to(dev).requires_grad_(True)
    pp = hip.pair_plan(pos_d, pl, cutoff)
    assert int(pp.count[0]) == full // 2
    out = hip.cfconv_pair(x1d, pd[0], pd[1], pd[2], pd[3], pp, gs.offset.to(dev), gs.coeff, cutoff)
    out2, Wf = hip.cfconv_pair_forward(x1d.detach(), pp, pd[0].detach(), pd[1].detach(), pd[2].detach(), pd[3].detach(),
                                       gs.offset.to(dev), gs.coeff, cutoff)
    scale = float(ref.abs().max())
    assert_close(out[:N], ref.detach(), 1e-4, 1e-4 * scale, "pair cfconv fwd")
    assert torch.equal(out, out2), "bitwise reproducible"
    if pad:
        assert float(out[N:].abs().max()) == 0.0
    # filter rows: pair (i < j) against the oracle's edge j -> i
    key = {(int(a), int(c)): k for k, (a, c) in enumerate(zip(ei[0].tolist(), ei[1].tolist()))}
    P2 = int(pp.count[0])
    pi, pj, pdist = pp.pi[:P2].cpu().tolist(), pp.pj[:P2].cpu().tolist(), pp.pd[:P2].cpu()
    rows = torch.zeros(P2, 128)
    for k, (a, c) in enumerate(zip(pi, pj)):
        assert a < c
        e = key.get((c, a))
        assert (e is None) == (float(pdist[k]) < 0), "pairs beyond the cutoff carry distance -1"
        if e is not None:
            rows[k] = Wf_ref[e].detach()
    assert_close(Wf[:P2], rows, 1e-4, 1e-5, "pair filter rows")
    (out * w_full.to(dev)).sum().backward()
    assert_close(x1d.grad[:N], x1.grad, 1e-4, 1e-4 * float(x1.grad.abs().max()), "pair g_x1")
    for name, a, r in zip(("gW1", "gb1", "gW2", "gb2"), pd, ps):
        assert_close(a.grad, r.grad, 1e-3, 2e-4 * float(r.grad.abs().max()), f"pair {name}")


@pytest.mark.parametrize("L,group", [(3, 3), (6, 6), (5, 2)])
def test_cfconv_pair_all_blocks_in_one_launch(dev, L, group):
    """Round 6: the filter rows of ALL interaction blocks from one launch (msde_cfconv_pair_filter_multi: they depend on the
    distances only, schnet.py:96-104,141-145) are the per-block launches' rows bit for bit, and the blocks' filter-network
    weight gradients collected by hip.CfBwdBatch (msde_cfconv_pair_bwd_w_multi: one launch per `group` blocks inside a
    parameter-gradient batch) equal the per-block launches' gradients (same kernels; the slab count per block differs, so the
    sums agree to rounding) -- including a group that does not divide the number of blocks."""
    from moleculesde_amd import hip, plan as P
    torch.manual_seed(11)
    b = _toy_graph(12, 40)
    pl = P.plan_to(P.build_plan(b), dev)
    N, G, cutoff = b.x.size(0), 51, 10.0
    gs = R.GaussianSmearing(0.0, cutoff, G)
    off = gs.offset.to(dev)
    pp = hip.pair_plan(b.positions.to(dev), pl, cutoff)
    nets = []
    for _ in range(L):
        blk = R.InteractionBlock(hidden_channels=64, num_gaussians=G, num_filters=128, cutoff=cutoff)
        with torch.no_grad():
            blk.mlp[0].bias.normal_(0, 0.1); blk.mlp[2].bias.normal_(0, 0.1)
        nets.append([q.detach().to(dev) for q in (blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight, blk.mlp[2].bias)])
    Wfs = hip.cfconv_pair_filters(pp, nets, off, gs.coeff, cutoff)
    xs = [torch.randn(N, 128, device=dev) for _ in range(L)]
    ws = [torch.randn(N, 128, device=dev) for _ in range(L)]
    P2 = int(pp.count[0])
    for l in range(L):
        _, Wf1 = hip.cfconv_pair_forward(xs[l], pp, *nets[l], off, gs.coeff, cutoff)
        assert torch.equal(Wfs[l][:P2], Wf1[:P2]), f"filter rows of block {l}"

    def run(batched):
        ps = [[q.clone().requires_grad_(True) for q in n] for n in nets]
        x = [t.clone().requires_grad_(True) for t in xs]
        bb = hip.CfBwdBatch(pp, off, gs.coeff, cutoff, L, group=group) if batched else None
        outs = [hip.cfconv_pair(x[l], *ps[l], pp, off, gs.coeff, cutoff, Wf=Wfs[l] if batched else None, bwd_batch=bb)
                for l in range(L)]
        loss = sum((o * w).sum() for o, w in zip(outs, ws))
        params = [q for n in ps for q in n]
        if batched:
            slabs.begin_param_grad_batch(params)
            try:
                loss.backward()
            finally:
                slabs.finish_param_grad_batch()
        else:
            loss.backward()
        torch.cuda.synchronize()
        return outs, x, ps
    o1, x1_, p1 = run(False)
    o2, x2_, p2 = run(True)
    for l in range(L):
        assert torch.equal(o1[l], o2[l]), "aggregation on precomputed rows"
        assert torch.equal(x1_[l].grad, x2_[l].grad), "input gradient"
        for name, a, r in zip(("gW1", "gb1", "gW2", "gb2"), p2[l], p1[l]):
            assert a.grad is not None and torch.isfinite(a.grad).all()
            assert_close(a.grad, r.grad, 1e-4, 1e-5 * float(r.grad.abs().max()), f"block {l} {name} (batched launch)")


def test_edge_geometry(dev):
    from moleculesde_amd import hip
    torch.manual_seed(6)
    b = _toy_graph(8, 16)
    N = b.x.size(0)
    pl = _plan_for(b.extended_edge_index, N, dev)
    pos = b.positions + 0.3 * torch.randn_like(b.positions)
    Wd, Wc = torch.randn(32), torch.randn(32)
    fd, fi, fj, ang, basis = hip.edge_geometry(pos.to(dev), pl, Wd.to(dev), Wc.to(dev))
    row, col = pl.src.cpu().long(), pl.dst.cpu().long()
    cd, cc, cv = R.coord2basis(pos, row, col)
    assert_close(basis, torch.cat([cd, cc, cv], dim=1), 1e-4, 1e-5, "basis")
    dist = (pos[row] - pos[col]).norm(dim=-1, keepdim=True)
    gf = R.GaussianFourierProjection(32)
    gf.W.data = Wd
    # sin/cos of arguments up to ~1e2 rad: absolute tolerance = |arg| * 2^-23 * few
    assert_close(fd, gf(dist), 0, 5e-5, "distance fourier")
    eb = torch.stack([cd, cc, cv], dim=1)
    ci = torch.matmul(eb, pos[row].unsqueeze(-1)).squeeze(-1)
    cj = torch.matmul(eb, pos[col].unsqueeze(-1)).squeeze(-1)
    ci[:, 1] = ci[:, 1].abs()
    cj[:, 1] = cj[:, 1].abs()
    gc = R.GaussianFourierProjection(32)
    gc.W.data = Wc
    emb = lambda c: torch.cat([gc(c[:, 0:1]), gc(c[:, 2:3])], dim=-1)
    assert_close(fi, emb(ci), 0, 2e-4, "frame fourier i")
    assert_close(fj, emb(cj), 0, 2e-4, "frame fourier j")
    pcos = (ci * cj).sum(-1, keepdim=True) / (ci.norm(dim=-1, keepdim=True) + 1e-6) / (cj.norm(dim=-1, keepdim=True) + 1e-6)
    assert_close(ang[:, 1:2], pcos, 1e-4, 1e-5, "pseudo cos")
    good = (pcos.abs() < 0.999).flatten()          # sqrt(1-c^2) is ill-conditioned near |c| = 1
    assert_close(ang[good, 0:1], torch.sqrt(1 - pcos[good] ** 2), 1e-3, 1e-4, "pseudo sin")


@pytest.mark.parametrize("H,Ch", [(8, 4), (4, 8), (2, 2)])
def test_edge_attention(dev, H, Ch):
    from moleculesde_amd import hip
    torch.manual_seed(10 + H)
    b = _toy_graph(11, 16)
    N = b.x.size(0)
    pl = _plan_for(b.extended_edge_index, N, dev)
    D = H * Ch
    src, dst = pl.src.cpu().long(), pl.dst.cpu().long()
    q, k, v = (torch.randn(N, D, requires_grad=True) for _ in range(3))
    ee = torch.randn(pl.E, D, requires_grad=True)
    qe = q[dst].view(-1, H, Ch)
    ke = k[src].view(-1, H, Ch) + ee.view(-1, H, Ch)
    alpha = R.segment_softmax((qe * ke).sum(-1) / math.sqrt(Ch), dst, N)
    ref = R.scatter_sum((v[src].view(-1, H, Ch) + ee.view(-1, H, Ch)) * alpha.unsqueeze(-1), dst, N).view(N, D)
    w = torch.randn_like(ref)
    (ref * w).sum().backward()
    ds = [t.detach().to(dev).requires_grad_(True) for t in (q, k, v, ee)]
    out = hip.edge_attention(*ds, pl, H, 0.0, 0)
    assert_close(out, ref, 1e-4, 1e-5, "attention fwd")
    (out * w.to(dev)).sum().backward()
    for name, a, r in zip("qkve", ds, (q, k, v, ee)):
        assert_close(a.grad, r.grad, 1e-3, 2e-5, f"attention g_{name}")


def test_edge_attention_dropout(dev):
    """Dropout on alpha (hard-wired p = 0.1 in the reference): mask is a pure function of (seed, edge,
    head) -> forward is reproducible, backward uses the same mask, and E[out] is preserved."""
    from moleculesde_amd import hip
    torch.manual_seed(12)
    b = _toy_graph(13, 64)
    N = b.x.size(0)
    pl = _plan_for(b.extended_edge_index, N, dev)
    q, k, v = (torch.randn(N, 32, device=dev) for _ in range(3))
    ee = torch.randn(pl.E, 32, device=dev)
    o0 = hip.edge_attention(q, k, v, ee, pl, 8, 0.0, 0)
    o1 = hip.edge_attention(q, k, v, ee, pl, 8, 0.1, 1234)
    o1b = hip.edge_attention(q, k, v, ee, pl, 8, 0.1, 1234)
    o2 = hip.edge_attention(q, k, v, ee, pl, 8, 0.1, 99)
    assert torch.equal(o1, o1b) and not torch.equal(o1, o2) and not torch.equal(o1, o0)
    acc = torch.zeros_like(o0)
    for s in range(200):
        acc += hip.edge_attention(q, k, v, ee, pl, 8, 0.1, 5000 + s)
    assert float((acc / 200 - o0).abs().mean()) < 0.05 * float(o0.abs().mean()) + 0.02
    # backward consistency under a fixed mask: directional finite difference on v
    vv = v.clone().requires_grad_(True)
    out = hip.edge_attention(q, k, vv, ee, pl, 8, 0.1, 77)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    dv = torch.randn_like(v)
    eps = 1e-2
    fp = (hip.edge_attention(q, k, v + eps * dv, ee, pl, 8, 0.1, 77) * w).sum()
    fm = (hip.edge_attention(q, k, v - eps * dv, ee, pl, 8, 0.1, 77) * w).sum()
    fd = float((fp - fm) / (2 * eps))
    an = float((vv.grad * dv).sum())
    assert abs(fd - an) <= 2e-3 * max(1.0, abs(an))     # out is linear in v: FD is exact up to rounding


def test_frame_mix_mean(dev):
    from moleculesde_amd import hip
    torch.manual_seed(14)
    b = _toy_graph(15, 16)
    N = b.x.size(0)
    pl = _plan_for(b.extended_edge_index, N, dev)
    dst = pl.dst.cpu().long()
    coff = torch.randn(pl.E, 3, requires_grad=True)
    basis = torch.randn(pl.E, 9)
    mix = coff[:, :1] * basis[:, 0:3] + coff[:, 1:2] * basis[:, 3:6] + coff[:, 2:3] * basis[:, 6:9]
    ref = R.scatter_mean(mix, dst, N)
    w = torch.randn_like(ref)
    (ref * w).sum().backward()
    cd = coff.detach().to(dev).requires_grad_(True)
    out = hip.frame_mix_mean(cd, basis.to(dev), pl)
    assert_close(out, ref, 1e-5, 1e-6, "frame mix fwd")
    (out * w.to(dev)).sum().backward()
    assert_close(cd.grad, coff.grad, 1e-5, 1e-6, "frame mix g_coff")


def test_segment_reduce_molecule(dev):
    from moleculesde_amd import hip, plan as P
    torch.manual_seed(16)
    b = _toy_graph(17, 12)
    pl = P.plan_to(P.build_plan(b), dev)
    x = torch.randn(b.x.size(0), 20, requires_grad=True)
    ref = R.scatter_mean(x, b.batch, b.num_graphs)
    w = torch.randn_like(ref)
    (ref * w).sum().backward()
    xd = x.detach().to(dev).requires_grad_(True)
    out = hip.segment_reduce(xd, pl.mol_ptr, pl.batch_i32, mean=True)
    assert_close(out, ref, 1e-5, 1e-6, "readout mean")
    (out * w.to(dev)).sum().backward()
    assert_close(xd.grad, x.grad, 1e-5, 1e-6, "readout grad")


@pytest.mark.parametrize("in_place", [False, True])
def test_adam_flat_matches_torch_adam(dev, in_place):
    """FlatAdam vs torch.optim.Adam over 25 steps, two lr groups, weight decay: the flat path (gather kernel +
    msde_adam_flat, used under data parallelism) and the chunk-table path that reads .grad in place.  One tensor
    spans several chunks, one is skipped by autograd in some steps (zero gradient by our definition)."""
    from moleculesde_amd.optim import FlatAdam
    torch.manual_seed(18)
    ps = [torch.nn.Parameter(torch.randn(37, 5)), torch.nn.Parameter(torch.randn(11)), torch.nn.Parameter(torch.randn(64, 3)),
          torch.nn.Parameter(torch.randn(70, 77))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt_ref = torch.optim.Adam([{"params": ref[:2], "lr": 1e-2}, {"params": ref[2:], "lr": 3e-3}], lr=1e-2, weight_decay=0.01)
    pd = [torch.nn.Parameter(p.detach().to(dev)) for p in ps]
    opt = FlatAdam([{"params": pd[:2], "lr": 1e-2}, {"params": pd[2:], "lr": 3e-3}], weight_decay=0.01)
    for step in range(25):
        gs = [torch.randn_like(p) for p in ps]
        for p, g in zip(ref, gs):
            p.grad = g.clone()
        if step % 5 == 4:
            ref[1].grad = torch.zeros_like(ref[1])
        opt_ref.step()
        for p, g in zip(pd, gs):
            p.grad = g.to(dev)
        if step % 5 == 4:                 # a parameter without a gradient this step == zero gradient
            pd[1].grad = None
        if in_place:
            opt.step_from_grads()
        else:
            opt.gather_grads()
            opt.step()
    for a, r in zip(pd, ref):
        assert_close(a, r, 1e-5, 1e-6, "adam params after 25 steps")


@pytest.mark.parametrize("M,N,K,bias", [(3588, 300, 300, True), (3588, 600, 300, True), (3588, 128, 300, False),
                                        (49090, 128, 51, True), (49090, 128, 128, True), (35186, 32, 300, True),
                                        (35186, 3, 128, True), (35186, 32, 66, True), (1000, 119, 728, True),
                                        (77, 5, 7, True), (1, 1, 1, False), (257, 65, 33, True)])
def test_linear_mfma_gemm(dev, M, N, K, bias):
    """fp32 MFMA Linear (fwd, dgrad, wgrad + bias grad, all three tile configs, ragged edges, the
    scalar-load path for K % 4 != 0) vs torch CPU fp64 accumulation of the same fp32 inputs."""
    from moleculesde_amd import hip
    hip.set_linear_mode("hip")           # exercise the hand-written kernels for all three products
    try:
        g = torch.Generator().manual_seed(M + N + K)
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / math.sqrt(K)
        b = torch.randn(N, generator=g) if bias else None
        gy = torch.randn(M, N, generator=g)
        xd = x.to(dev).requires_grad_(True)
        wd = w.to(dev).requires_grad_(True)
        bd = b.to(dev).requires_grad_(True) if bias else None
        y = hip.linear(xd, wd, bd)
        y.backward(gy.to(dev))
        x64, w64 = x.double(), w.double()
        yr = x64 @ w64.t() + (b.double() if bias else 0)
        # fp32 fma chain of length K: error ~ sqrt(K) * 2^-24 * |x||w|
        assert_close(y, yr, 2e-6 * math.sqrt(K) + 1e-6, 1e-5, "linear fwd")
        assert_close(xd.grad, gy.double() @ w64, 2e-6 * math.sqrt(N) + 1e-6, 1e-5 * math.sqrt(N), "linear dgrad")
        gwr = gy.double().t() @ x64
        assert_close(wd.grad, gwr, 1e-5, 2e-6 * math.sqrt(M) * 4, "linear wgrad")
        if bias:
            assert_close(bd.grad, gy.double().sum(0), 1e-5, 2e-6 * math.sqrt(M) * 4, "linear bias grad")
        # reproducible: the split-M slabs are reduced in a fixed order
        xd2 = x.to(dev).requires_grad_(True)
        wd2 = w.to(dev).requires_grad_(True)
        hip.linear(xd2, wd2, bd.detach() if bias else None).backward(gy.to(dev))
        assert torch.equal(wd2.grad, wd.grad) and torch.equal(xd2.grad, xd.grad)
    finally:
        hip.set_linear_mode("auto")
    xd3 = x.to(dev).requires_grad_(True)
    wd3 = w.to(dev).requires_grad_(True)
    y3 = hip.linear(xd3, wd3, bd.detach() if bias else None)
    y3.backward(gy.to(dev))
    assert_close(y3, yr, 1e-4, 1e-4, "auto-dispatch fwd")
    assert_close(wd3.grad, gwr, 1e-4, 2e-6 * math.sqrt(M) * 8, "auto-dispatch wgrad")


@pytest.mark.parametrize("M,C,relu", [(3588, 600, True), (3588, 300, False), (35186, 300, True), (7, 5, True), (1, 3, False),
                                      (300, 64, True)])
def test_batchnorm_train_kernels(dev, M, C, relu):
    """csrc/norm.hip vs torch CPU BatchNorm1d (training mode) + ReLU: output, input/affine gradients and
    the running-statistics update."""
    from moleculesde_amd import hip
    torch.manual_seed(M + C)
    x = (torch.randn(M, C) * 2 + 0.5).requires_grad_(True)
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.normal_(1, 0.3); bn.bias.normal_(0, 0.3)
    if M == 1:
        pytest.skip("torch refuses a single row in training mode")
    y_pre = bn(x)
    y = torch.relu(y_pre) if relu else y_pre
    w = torch.randn(M, C)
    (y * w).sum().backward()
    xd = x.detach().to(dev).requires_grad_(True)
    gd = bn.weight.detach().clone().to(dev).requires_grad_(True)
    bd = bn.bias.detach().clone().to(dev).requires_grad_(True)
    rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
    yd = hip.batch_norm_train(xd, gd, bd, rm, rv, 1e-5, 0.1, relu)
    assert_close(yd, y.detach(), 1e-4, 1e-5, "bn fwd")
    (yd * w.to(dev)).sum().backward()
    # a pre-activation within rounding of 0 may fall on either side of the fused ReLU gate: those few
    # elements are excluded from the element-wise input-gradient check (the column sums below keep them)
    sure = (y_pre.detach().abs() > 1e-5) if relu else torch.ones_like(y_pre, dtype=torch.bool)
    assert int((~sure).sum()) <= 1e-4 * sure.numel() + 2
    assert_close(torch.where(sure.to(dev), xd.grad, torch.zeros_like(xd.grad)), torch.where(sure, x.grad, torch.zeros_like(x.grad)),
                 1e-3, 1e-5 * float(x.grad.abs().max()) + 1e-6, "bn dx")
    # column sums: an element on the gate boundary contributes |w| (dbeta) / |w xhat| (dgamma) or nothing
    unsure = (~sure).float()
    xhat = (y_pre.detach() - bn.bias.detach()) / bn.weight.detach()
    for got, ref, allow, what in ((gd.grad, bn.weight.grad, (unsure * (w * xhat).abs()).sum(0), "bn dgamma"),
                                  (bd.grad, bn.bias.grad, (unsure * w.abs()).sum(0), "bn dbeta")):
        err = (got.cpu() - ref).abs()
        bound = 1e-4 * ref.abs() + 1e-4 * float(ref.abs().max()) + 1e-6 + 1.01 * allow
        assert bool((err <= bound).all()), f"{what}: max excess {(err - bound).max().item():.3e}"
    assert_close(rm, bn.running_mean, 1e-5, 1e-6, "running mean")
    assert_close(rv, bn.running_var, 1e-4, 1e-6, "running var")


@pytest.mark.parametrize("N,D", [(3588, 300), (50, 16), (7, 5)])
def test_contrastive_ebm_kernels(dev, N, D):
    """Fused dual EBM-NCE loss (csrc/contrastive.hip) vs the oracle's dual_CL with the same permutations."""
    from moleculesde_amd import hip
    g = torch.Generator().manual_seed(N + D)
    X = (torch.randn(N, D, generator=g) * 0.3).requires_grad_(True)
    Y = (torch.randn(N, D, generator=g) * 0.3).requires_grad_(True)
    p1, p2 = torch.randperm(N, generator=g), torch.randperm(N, generator=g)
    ref, acc_ref = R.dual_CL(X, Y, 0.1, p1, p2)
    (ref * 1.7).backward()
    Xd = X.detach().to(dev).requires_grad_(True)
    Yd = Y.detach().to(dev).requires_grad_(True)
    loss, acc = hip.contrastive_ebm(Xd, Yd, p1.to(dev), p2.to(dev), 0.1)
    assert_close(loss, ref.detach(), 1e-5, 1e-6, "CL loss")
    assert abs(float(acc) - acc_ref) < 1e-6
    (loss * 1.7).backward()
    assert_close(Xd.grad, X.grad, 1e-4, 1e-5 * float(X.grad.abs().max()), "CL gX")
    assert_close(Yd.grad, Y.grad, 1e-4, 1e-5 * float(Y.grad.abs().max()), "CL gY")


@pytest.mark.parametrize("N,D,with_res", [(3588, 32, True), (100, 300, True), (5, 8, False), (777, 64, True)])
def test_res_layernorm_kernels(dev, N, D, with_res):
    from moleculesde_amd import hip
    torch.manual_seed(N + D)
    x = torch.randn(N, D, requires_grad=True)
    res = torch.randn(N, D, requires_grad=True) if with_res else None
    ln = torch.nn.LayerNorm(D)
    with torch.no_grad():
        ln.weight.normal_(1, 0.3); ln.bias.normal_(0, 0.3)
    y = ln(x) + (res if with_res else 0)
    w = torch.randn(N, D)
    (y * w).sum().backward()
    xd = x.detach().to(dev).requires_grad_(True)
    rd = res.detach().to(dev).requires_grad_(True) if with_res else None
    gd = ln.weight.detach().to(dev).requires_grad_(True)
    bd = ln.bias.detach().to(dev).requires_grad_(True)
    yd = hip.res_layernorm(xd, rd, gd, bd, ln.eps)
    assert_close(yd, y.detach(), 1e-4, 1e-5, "res+LN fwd")
    (yd * w.to(dev)).sum().backward()
    assert_close(xd.grad, x.grad, 1e-3, 1e-5, "LN gx")
    if with_res:
        assert_close(rd.grad, res.grad, 0, 0, "LN g_res")
    assert_close(gd.grad, ln.weight.grad, 1e-4, 1e-4 * float(ln.weight.grad.abs().max()), "LN ggamma")
    assert_close(bd.grad, ln.bias.grad, 1e-4, 1e-4 * float(ln.bias.grad.abs().max()), "LN gbeta")


# ------------------------------------------------------------------------------------------------
# pointwise stages (csrc/pointwise.hip)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 7, 1024, 3588 * 300])
def test_shifted_softplus_kernel(dev, n):
    from moleculesde_amd import hip
    torch.manual_seed(n)
    x = (torch.randn(n) * 6).requires_grad_(True)
    with torch.no_grad():
        x[: min(n, 3)] = torch.tensor([25.0, -30.0, 0.0])[: min(n, 3)]     # threshold branch, deep negative, zero
    ref = torch.nn.functional.softplus(x) - math.log(2.0)
    w = torch.randn(n)
    (ref * w).sum().backward()
    xd = x.detach().to(dev).requires_grad_(True)
    y = hip.shifted_softplus(xd)
    assert_close(y, ref.detach(), 1e-6, 1e-6, "ssp fwd")
    (y * w.to(dev)).sum().backward()
    assert_close(xd.grad, x.grad, 1e-5, 1e-6, "ssp bwd")


def test_silu_dropout_kernel(dev):
    from moleculesde_amd import hip
    torch.manual_seed(3)
    x = (torch.randn(3588, 32) * 2).requires_grad_(True)
    w = torch.randn(3588, 32)
    ref = torch.nn.functional.silu(x)
    (ref * w).sum().backward()
    xd = x.detach().to(dev).requires_grad_(True)
    y = hip.silu_dropout(xd)                                   # p = 0: plain SiLU
    assert_close(y, ref.detach(), 1e-6, 1e-6, "silu fwd")
    (y * w.to(dev)).sum().backward()
    assert_close(xd.grad, x.grad, 1e-5, 1e-6, "silu bwd")
    # p = 0.1: kept entries are silu/(1-p), dropped are 0, the backward uses the same mask, a new seed (or a new
    # device counter value) gives a new mask, the drop rate is right
    p = 0.1
    xd2 = x.detach().to(dev).requires_grad_(True)
    ctr = torch.zeros(1, dtype=torch.int64, device=dev)
    y1 = hip.silu_dropout(xd2, p, 1234, ctr)
    keep = y1 != 0
    rate = 1.0 - keep.float().mean().item()
    assert abs(rate - p) < 0.01, rate
    assert_close(torch.where(keep, y1, torch.zeros_like(y1)), torch.where(keep.cpu(), ref.detach() / (1 - p), torch.zeros_like(ref)),
                 1e-5, 1e-6, "dropout kept values")
    (y1 * w.to(dev)).sum().backward()
    gref = torch.where(keep.cpu(), x.grad / (1 - p), torch.zeros_like(x.grad))
    assert_close(xd2.grad, gref, 1e-5, 1e-6, "dropout bwd mask")
    y2 = hip.silu_dropout(xd2, p, 1234, ctr)
    assert torch.equal(y1, y2)                                # same (seed, counter) -> same mask
    ctr.add_(1)
    y3 = hip.silu_dropout(xd2, p, 1234, ctr)
    assert not torch.equal(y1 != 0, y3 != 0)                  # replay with an advanced counter -> fresh mask
    assert not torch.equal(y1 != 0, hip.silu_dropout(xd2, p, 99, None) != 0)


def test_mul_add_kernel(dev):
    from moleculesde_amd import hip
    torch.manual_seed(4)
    a, b, c = (torch.randn(35186, 32).requires_grad_(True) for _ in range(3))
    w = torch.randn(35186, 32)
    ((a * b + c) * w).sum().backward()
    ad, bd, cd = (t.detach().to(dev).requires_grad_(True) for t in (a, b, c))
    out = hip.mul_add(ad, bd, cd)
    assert_close(out, (a * b + c).detach(), 0, 0, "mul_add fwd")          # same two roundings as the reference
    (out * w.to(dev)).sum().backward()
    for got, ref, what in ((ad.grad, a.grad, "ga"), (bd.grad, b.grad, "gb"), (cd.grad, c.grad, "gc")):
        assert_close(got, ref, 0, 0, what)


@pytest.mark.parametrize("power", [0.0, 2.0])
def test_ve_position_loss_kernel(dev, power):
    """Fused VE position loss vs the reference's operator chain (SDE_model_2D_to_3D.py:425-432)."""
    from moleculesde_amd import hip, plan as P
    b = _toy_graph(9, 37)
    pl = P.plan_to(P.build_plan(b), dev)
    N = b.x.size(0)
    torch.manual_seed(11)
    scores = torch.randn(N, 3, requires_grad=True)
    noise = torch.randn(N, 3)
    std = torch.rand(N) + 0.3
    if power == 0:
        lp = torch.sum((scores - noise) ** 2, -1)
    else:
        lp = torch.sum((scores - noise) ** 2 * (std ** power).unsqueeze(1), -1)
    ref = R.scatter_mean(lp.unsqueeze(1), b.batch, int(b.num_graphs)).mean()
    (ref * 1.3).backward()
    sd = scores.detach().to(dev).requires_grad_(True)
    loss = hip.ve_position_loss(sd, noise.to(dev), std.to(dev), power, pl.mol_ptr, pl.batch_i32)
    assert_close(loss, ref.detach(), 1e-5, 1e-6, "VE position loss")
    (loss * 1.3).backward()
    assert_close(sd.grad, scores.grad, 1e-5, 1e-7, "VE position loss grad")


def test_ve_perturb_kernel(dev):
    """Fused VE perturbation vs the reference's operator chain (SDE_model_2D_to_3D.py:401-412)."""
    from moleculesde_amd import hip, plan as P
    from moleculesde_amd.geom3d.sde import VESDE
    b = _toy_graph(13, 37)
    pl = P.plan_to(P.build_plan(b), dev)
    B, T, eps = int(b.num_graphs), 1000, 1e-6
    g = torch.Generator().manual_seed(5)
    noise = torch.randn(b.positions.shape, generator=g)
    draws = torch.randint(0, T, (B // 2 + 1,), generator=g)
    sde = VESDE(sigma_min=0.2, sigma_max=1.0, N=T)
    time_step = torch.cat([draws, T - draws - 1], dim=0)[:B]
    t = time_step / T * (1 - eps) + eps
    t_pos = t.index_select(0, b.batch)
    mean, std = sde.marGINal_prob(b.positions, t_pos)
    ref = mean + std[:, None] * noise
    out, sd = hip.ve_perturb(b.positions.to(dev), noise.to(dev), draws.to(dev), pl.batch_i32, B, T, eps, 0.2, 1.0)
    assert_close(sd, std, 1e-6, 1e-7, "std per atom")
    assert_close(out, ref, 1e-6, 1e-6, "perturbed positions")


def test_batched_slab_reduction_matches_immediate(dev):
    """Weight/bias gradients with the slab reductions of several layers batched into one launch
    (hip.begin/finish_param_grad_batch) are bit-identical to the per-layer reduction... up to the different
    (but fixed) lane interleave of the two reduction kernels: compared at fp32 tolerance, and exactly against a
    second batched run (determinism)."""
    from moleculesde_amd import hip
    torch.manual_seed(21)
    shapes = [(3588, 300, 300, True), (35186, 32, 32, True), (3588, 3, 128, True), (35186, 128, 64, False), (300, 7, 5, True)]
    layers = []
    for M, N, K, bias in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev, requires_grad=True)
        b = torch.randn(N, device=dev, requires_grad=True) if bias else None
        g = torch.randn(M, N, device=dev)
        layers.append((x, w, b, g))

    def run(batched):
        for x, w, b, g in layers:
            w.grad = None
            if b is not None:
                b.grad = None
        if batched:
            slabs.begin_param_grad_batch()
        for x, w, b, g in layers:
            hip.linear(x, w, b).backward(g)
        if batched:
            slabs.finish_param_grad_batch()
        torch.cuda.synchronize()
        return [(w.grad.clone(), None if b is None else b.grad.clone()) for x, w, b, g in layers]

    ref = run(False)
    got = run(True)
    again = run(True)
    for (M, N, K, bias), (gw0, gb0), (gw1, gb1), (gw2, gb2) in zip(shapes, ref, got, again):
        tol = 2e-6 * math.sqrt(M) * 8
        assert_close(gw1, gw0.cpu(), 1e-4, tol, f"batched wgrad {M}x{N}x{K}")
        assert torch.equal(gw1, gw2)
        if bias:
            assert_close(gb1, gb0.cpu(), 1e-4, tol, f"batched bias grad {M}x{N}x{K}")
            assert torch.equal(gb1, gb2)


@pytest.mark.parametrize("E,H", [(35186, 32), (1000, 16), (3, 32)])
def test_frame_mlp_and_pair_linear_fused_operators(dev, E, H):
    """hip._FrameMLP (project(cat([angle, coff_mlp(feat_i), coff_mlp(feat_j)])), SDE_model_2D_to_3D.py:366-370) and
    hip._PairLinear (edge_2D_emb[0] on cat(h[row], h[col]) as one per-node product, :386-388): values and every
    parameter gradient against torch autograd in fp64 -- with the weight gradients formed on the spot, and queued in a
    parameter-gradient batch (2-D rows of msde_reduce_slabs_multi write column blocks of the wider gradients); the cached
    re-laid-out weight copies (wcache.weight_layout) follow an in-place parameter update."""
    from moleculesde_amd import hip
    g = torch.Generator().manual_seed(E + H)
    mk = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    feat_i, feat_j, angle = mk(E, 4 * H), mk(E, 4 * H), mk(E, 2)
    Wc, bc = mk(H, 4 * H, sc=(4 * H) ** -0.5), mk(H, sc=0.1)
    W1, b1 = mk(H, 2 * H + 2, sc=(2 * H) ** -0.5), mk(H, sc=0.1)
    W2, b2 = mk(H, H, sc=H ** -0.5), mk(H, sc=0.1)
    g_out = mk(E, H)

    def ref_run(scale):
        P = [t.double().clone().requires_grad_(True) for t in (Wc, bc, W1, b1, W2, b2)]
        with torch.no_grad():
            P[2].mul_(scale)
        Wc_, bc_, W1_, b1_, W2_, b2_ = P
        ei, ej = feat_i.double() @ Wc_.t() + bc_, feat_j.double() @ Wc_.t() + bc_
        z = torch.cat([angle.double(), ei, ej], dim=1) @ W1_.t() + b1_
        out = torch.nn.functional.silu(z) @ W2_.t() + b2_
        return out, torch.autograd.grad(out, P, g_out.double())

    feat = torch.stack([feat_i, feat_j], dim=1).reshape(2 * E, 4 * H).to(dev)
    X = torch.full((2 * E, H + 4), float("nan"), device=dev)
    X[0::2, H:H + 2] = angle.to(dev); X[0::2, H + 2:] = 0; X[1::2, H:] = 0
    P = [t.to(dev).clone().requires_grad_(True) for t in (Wc, bc, W1, b1, W2, b2)]
    for batched, scale in ((False, 1.0), (True, 1.0), (True, 0.5)):
        if scale != 1.0:
            with torch.no_grad():
                P[2].mul_(scale)            # in-place update: the cached permuted copy of W1 must follow
        out_ref, g_ref = ref_run(scale)
        for p in P:
            p.grad = None
        if batched:
            slabs.begin_param_grad_batch(P)
        out = hip._FrameMLP.apply(feat, X.clone(), *P)
        out.backward(g_out.to(dev))
        if batched:
            slabs.finish_param_grad_batch()
        assert_close(out, out_ref, 2e-5, 2e-5, f"frame mlp out (batched={batched})")
        for name, p, r in zip(("Wc", "bc", "W1", "b1", "W2", "b2"), P, g_ref):
            assert_close(p.grad, r, 2e-4, 2e-4 * max(1.0, float(r.abs().max())), f"frame mlp grad {name} (batched={batched}, scale={scale})")
    # pair linear
    M, D = 777, 4 * H
    h, W, b, gAB = mk(M, D), mk(D, 2 * D, sc=(2 * D) ** -0.5), mk(D, sc=0.1), mk(M, 2 * D)
    R = [t.double().clone().requires_grad_(True) for t in (h, W, b)]
    AB_ref = torch.cat([R[0] @ R[1][:, :D].t(), R[0] @ R[1][:, D:].t() + R[2]], dim=1)
    g_ref = torch.autograd.grad(AB_ref, R, gAB.double())
    Q = [t.to(dev).clone().requires_grad_(True) for t in (h, W, b)]
    for batched in (False, True):
        for q in Q:
            q.grad = None
        if batched:
            slabs.begin_param_grad_batch(Q[1:])
        AB = hip._PairLinear.apply(*Q)
        AB.backward(gAB.to(dev))
        if batched:
            slabs.finish_param_grad_batch()
        assert_close(AB, AB_ref, 2e-5, 2e-5, "pair linear out")
        for name, q, r in zip(("h", "W", "b"), Q, g_ref):
            assert_close(q.grad, r, 2e-4, 2e-4 * max(1.0, float(r.abs().max())), f"pair linear grad {name} (batched={batched})")


def test_weight_copies_follow_parameter_edits_and_moves(dev):
    """The cached [K][N] copies the forward products read (wcache.weight_t): an in-place edit (version counter), an update
    through raw pointers (wcache.bump_weight_epoch, what FlatAdam does) and a parameter whose storage is re-pointed (what
    building a flat-buffer optimiser after a first forward does) are all seen by the next forward; the batched refresh
    drops the copy of the old storage instead of reading it."""
    from moleculesde_amd import hip
    torch.manual_seed(3)
    x = torch.randn(500, 64, device=dev)
    lin = torch.nn.Linear(64, 48).to(dev)
    ref = lambda: x.double() @ lin.weight.detach().double().t() + lin.bias.detach().double()
    with torch.no_grad():
        assert_close(hip.linear(x, lin.weight, lin.bias), ref(), 1e-5, 1e-5, "first forward")
        lin.weight.mul_(0.5)                                        # version counter moves
        assert_close(hip.linear(x, lin.weight, lin.bias), ref(), 1e-5, 1e-5, "after an in-place edit")
        lin.weight.data.view(-1)[::7] += 1.0                        # .data edit: no version bump ...
        wcache.bump_weight_epoch()                                     # ... the optimiser's contract
        assert_close(hip.linear(x, lin.weight, lin.bias), ref(), 1e-5, 1e-5, "after a raw-pointer update")
        lin.weight.data = (lin.weight.data * 2.0).clone()           # storage re-pointed
        wcache.refresh_weight_t()                                      # must not touch the old storage's entry
        assert_close(hip.linear(x, lin.weight, lin.bias), ref(), 1e-5, 1e-5, "after the storage moved")
        wcache.refresh_weight_t()
        torch.cuda.synchronize()


def test_weight_copy_refresh_captured_in_a_graph_survives_later_entries(dev):
    """a captured step holds the raw addresses of the refresh launch's tables and of every copy it writes.  Entries
    that appear (another model's first forward) or die (a temporary module) after the capture rebuild the EAGER tables;
    the captured launch keeps reading its own table and writing its own buffers, which are never handed back to the
    allocator.  After the replay the copy of the captured entry is fresh, the later entry is marked stale."""
    from moleculesde_amd import hip
    torch.manual_seed(4)
    lin_a = torch.nn.Linear(96, 80).to(dev)
    with torch.no_grad():
        wa = wcache.weight_t(lin_a.weight)
        assert torch.equal(wa, lin_a.weight.t())
        slabs.note_capture()
        wcache.refresh_weight_t()                                        # (eager: builds the tables a capture may not upload)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            keys = wcache.refresh_weight_t()
        assert len(keys) >= 1
        lin_b = torch.nn.Linear(64, 48).to(dev)                       # a later entry: the eager tables are rebuilt
        wb = wcache.weight_t(lin_b.weight)
        tmp = torch.nn.Linear(200, 120).to(dev)
        wcache.weight_t(tmp.weight)
        del tmp                                                       # ... and one that dies: its buffer must not be reused
        import gc
        gc.collect()
        junk = [torch.full((n,), float("nan"), device=dev) for n in (64, 512, 4096, 24000, 200 * 120)]   # grabs freed blocks
        lin_a.weight.data.add_(1.0)                                   # an update Python does not see (what Adam in a graph does)
        lin_b.weight.data.add_(2.0)
        g.replay()
        wcache.weight_copies_after_replay(keys)
        torch.cuda.synchronize()
        assert torch.equal(wa, lin_a.weight.t()), "the captured refresh wrote the copy it was captured with"
        assert all(bool(torch.isnan(j).all()) for j in junk), "a replay wrote into memory it no longer owns"
        assert torch.equal(wcache.weight_t(lin_b.weight), lin_b.weight.t()), "the later entry was marked stale and refreshed"
        del wb


def test_ve_perturb_rng_kernel(dev):
    """msde_ve_perturb_rng: the VE perturbation (SDE_model_2D_to_3D.py:401-412) with the draws made in the kernel.  The
    noise is N(0,1) (moments over 3 x 40k draws), pos_out = pos + std * noise, the time steps of molecule b and
    b + H are antithetic (ts + ts' = T - 1 => std * std' is one constant), every molecule's atoms share one std,
    reproducible for (seed, counter) and different for another counter value."""
    from moleculesde_amd import hip
    B, per, T = 256, 157, 1000
    N = B * per
    pos = torch.randn(N, 3, device=dev)
    batch = torch.arange(B, device=dev, dtype=torch.int32).repeat_interleave(per)
    ctr = torch.zeros(1, dtype=torch.int64, device=dev)
    smin, smax, eps = 0.1, 1.0, 1e-5
    noise, out, std = hip.ve_perturb_rng(pos, batch, B, T, eps, smin, smax, 1234, ctr)
    assert abs(float(noise.mean())) < 0.01 and abs(float(noise.var()) - 1.0) < 0.02
    assert abs(float((noise ** 3).mean())) < 0.03 and abs(float((noise ** 4).mean()) - 3.0) < 0.1
    assert_close(out, pos.double() + std.double()[:, None] * noise.double(), 1e-6, 1e-6, "pos + std * noise")     # (fused multiply-add in the kernel)
    sm = std.view(B, per)
    assert torch.equal(sm, sm[:, :1].expand(B, per))
    H = B // 2 + 1
    prod = sm[:B - H, 0].double() * sm[H:, 0].double()
    want = smin * smin * (smax / smin) ** (((T - 1) / T) * (1 - eps) + 2 * eps)
    assert_close(prod, torch.full_like(prod, want), 1e-5, 0, "antithetic time steps")
    ts = torch.round(((torch.log(sm[:, 0].double() / smin) / math.log(smax / smin)) - eps) / (1 - eps) * T)
    assert int(ts.min()) >= 0 and int(ts.max()) <= T - 1 and len(torch.unique(ts[:H])) > H // 2
    again = hip.ve_perturb_rng(pos, batch, B, T, eps, smin, smax, 1234, ctr)
    assert all(torch.equal(a, b) for a, b in zip((noise, out, std), again))
    ctr.add_(1)
    other = hip.ve_perturb_rng(pos, batch, B, T, eps, smin, smax, 1234, ctr)
    assert not torch.equal(other[0], noise) and not torch.equal(other[2], std)


@pytest.mark.parametrize("n", [1, 2, 37, 3588, 4096])
def test_randperm_kernel(dev, n):
    """msde_randperm: always a permutation; reproducible for a (seed, counter); a new seed or counter value gives
    another permutation; no position bias (mean image of every quarter of the range stays near the centre)."""
    from moleculesde_amd import hip
    ctr = torch.zeros(1, dtype=torch.int64, device=dev)
    p1 = hip.randperm(n, dev, 123, ctr)
    assert p1.dtype == torch.int32
    assert torch.equal(torch.sort(p1.long())[0].cpu(), torch.arange(n))
    assert torch.equal(p1, hip.randperm(n, dev, 123, ctr))
    if n >= 37:
        assert not torch.equal(p1, hip.randperm(n, dev, 124, ctr))
        ctr.add_(1)
        assert not torch.equal(p1, hip.randperm(n, dev, 123, ctr))
    pair = hip.randperm(n, dev, 77, None, count=2)            # two independent permutations from one launch
    assert pair.shape == (2, n)
    for r in range(2):
        assert torch.equal(torch.sort(pair[r].long())[0].cpu(), torch.arange(n))
    if n >= 37:
        assert not torch.equal(pair[0], pair[1])
    if n >= 3588:
        acc = torch.zeros(4, dtype=torch.float64)
        reps = 64
        for r in range(reps):
            p = hip.randperm(n, dev, 1000 + r, None).double().cpu()
            acc += p.view(4, -1).mean(1) if n % 4 == 0 else torch.stack([c.mean() for c in p.chunk(4)])
        acc /= reps
        sigma = n / math.sqrt(12.0) / math.sqrt(reps * (n // 4))
        assert bool(((acc - (n - 1) / 2).abs() < 6 * sigma).all()), acc


@pytest.mark.parametrize("silu_out", [False, True])
def test_gat_tail_kernel(dev, silu_out):
    """Fused GATLayer tail (csrc/gat_tail.hip) vs the reference's operator chain on the CPU (p = 0), all ten
    gradients; with p = 0.1 vs the kernel-per-stage HIP path, which draws the identical dropout mask."""
    from moleculesde_amd import hip
    from moleculesde_amd.geom3d import nn as _nn
    torch.manual_seed(31)
    N, D = 3588, 32
    n1, n2 = torch.nn.LayerNorm(D), torch.nn.LayerNorm(D)
    f0, f3 = torch.nn.Linear(D, D), torch.nn.Linear(D, D)
    with torch.no_grad():
        for m in (n1, n2):
            m.weight.normal_(1, 0.3); m.bias.normal_(0, 0.3)
    x = torch.randn(N, D, requires_grad=True)
    res = torch.randn(N, D, requires_grad=True)
    w = torch.randn(N, D)
    y1 = res + n1(x)
    out = y1 + n2(f3(torch.nn.functional.silu(f0(y1))))
    out = torch.nn.functional.silu(out) if silu_out else out
    (out * w).sum().backward()
    params = [n1.weight, n1.bias, f0.weight, f0.bias, f3.weight, f3.bias, n2.weight, n2.bias]
    ref_grads = [x.grad, res.grad] + [p.grad for p in params]

    def dev_modules():
        m1, m2 = torch.nn.LayerNorm(D).to(dev), torch.nn.LayerNorm(D).to(dev)
        g0, g3 = _nn.Linear(D, D).to(dev), _nn.Linear(D, D).to(dev)
        with torch.no_grad():
            for a, b in ((m1, n1), (m2, n2), (g0, f0), (g3, f3)):
                a.weight.copy_(b.weight); a.bias.copy_(b.bias)
        return m1, g0, g3, m2

    m1, g0, g3, m2 = dev_modules()
    xd, rd = x.detach().to(dev).requires_grad_(True), res.detach().to(dev).requires_grad_(True)
    od = hip.gat_tail(xd, rd, m1, g0, g3, m2, 0.0, 7, None, silu_out)
    assert_close(od, out.detach(), 1e-4, 1e-5, "gat tail fwd")
    (od * w.to(dev)).sum().backward()
    got = [xd.grad, rd.grad, m1.weight.grad, m1.bias.grad, g0.weight.grad, g0.bias.grad, g3.weight.grad, g3.bias.grad,
           m2.weight.grad, m2.bias.grad]
    names = ["g_x", "g_res", "ln1_g", "ln1_b", "W0", "b0", "W3", "b3", "ln2_g", "ln2_b"]
    for a, b, nme in zip(got, ref_grads, names):
        assert_close(a, b, 2e-4, 2e-5 * float(b.abs().max()) + 1e-6, f"gat tail grad {nme}")

    # dropout: same (seed, index) mask as the staged path
    p, seed = 0.1, 1234
    ctr = torch.full((1,), 5, dtype=torch.int64, device=dev)
    outs = []
    for fused in (True, False):
        m1, g0, g3, m2 = dev_modules()
        xd, rd = x.detach().to(dev).requires_grad_(True), res.detach().to(dev).requires_grad_(True)
        if fused:
            o = hip.gat_tail(xd, rd, m1, g0, g3, m2, p, seed, ctr, silu_out)
        else:
            y1d = hip.res_layernorm(xd, rd, m1.weight, m1.bias, m1.eps)
            o = hip.res_layernorm(g3(hip.silu_dropout(g0(y1d), p, seed, ctr)), y1d, m2.weight, m2.bias, m2.eps)
            o = hip.silu_dropout(o) if silu_out else o
        (o * w.to(dev)).sum().backward()
        outs.append([o.detach(), xd.grad, rd.grad, g0.weight.grad, g3.bias.grad, m2.weight.grad])
    for a, b, nme in zip(outs[0], outs[1], ["out", "g_x", "g_res", "W0", "b3", "ln2_g"]):
        assert_close(a, b.cpu(), 2e-4, 2e-5 * float(b.abs().max()) + 1e-6, f"gat tail with dropout: {nme}")


# ------------------------------------------------------------------ general fused GEMM (gemm_ex.hip) ---
def _act_ref(name):
    import torch.nn.functional as F
    return {None: lambda z: z, "tanh": torch.tanh, "silu": F.silu, "elu": F.elu, "relu": F.relu,
            "ssp": lambda z: F.softplus(z) - 0.6931471805599453}[name]


@pytest.mark.parametrize("M,N,K", [(3588, 728, 364), (777, 119, 728), (100, 30, 30), (52680, 60, 32), (5, 300, 300),
                                   (1, 1, 7), (300, 176, 300)])
@pytest.mark.parametrize("km", [False, True])
def test_gemm_ex_plain_and_kmajor(dev, M, N, K, km):
    """C = A B^T + bias with SiLU on a column range and the pre-activation stored, against fp64 torch; both B
    layouts, aligned and unaligned shapes (scalar-load path), M tails, K tails."""
    from moleculesde_amd import hip
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    Bop = W.t().contiguous() if km else W
    out = torch.full((M, N), float("nan"), device=dev)
    Z = torch.full((M, N), float("nan"), device=dev)
    lo, hi = N // 4, N
    hip.gemm_ex(A, Bop, out, bias=b, act="silu", act_cols=(lo, hi), Z=Z, b_kmajor=km)
    zr = A.double() @ W.double().t() + b.double()
    ref = zr.clone()
    ref[:, lo:hi] = torch.nn.functional.silu(zr[:, lo:hi])
    assert_close(Z, zr, 1e-5, 2e-5, "pre-activation")
    assert_close(out, ref, 1e-5, 2e-5, "gemm_ex out")


@pytest.mark.parametrize("M,N,K", [(21, 300, 300), (21, 128, 300), (420, 128, 51), (420, 128, 128), (420, 51, 128), (1, 1, 1),
                                   (33, 35, 17), (512, 300, 512), (42, 1, 300), (7, 3, 16), (64, 64, 500),
                                   (3588, 16, 364), (3588, 16, 512), (3712, 32, 128), (8192, 32, 300), (3588, 5, 119)])
@pytest.mark.parametrize("km", [False, True])
@pytest.mark.parametrize("bias", [False, True])
def test_gemm_ex_small_problem_kernel(dev, M, N, K, km, bias):
    """Plain products of <= 512 rows (or <= 32 output columns and <= 8192 rows: the dense head's skinny node-level products,
    invariant_scorenetwork_dense.py:118-131) and K <= 512 (the MD17 step's ~130 per step, finetune_MD17.py:47-78) run on
    gemm_small_kernel (four waves split K, operands straight into MFMA registers): against fp64 torch; both B layouts,
    unaligned K / leading dimensions, edges of every kind, operands that are column blocks of wider buffers."""
    from moleculesde_amd import hip
    g = torch.Generator().manual_seed(M * 7 + N + K)
    wide = torch.randn(M, K + 8, generator=g).to(dev)
    A = wide[:, 4:4 + K] if K % 4 == 0 else wide[:, 3:3 + K]          # row stride != K; aligned only in the first case
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev) if bias else None
    Bop = W.t().contiguous() if km else W
    out_w = torch.full((M, N + 5), float("nan"), device=dev)
    out = out_w[:, 2:2 + N]
    hip.gemm_ex(A, Bop, out, bias=b, b_kmajor=km)
    ref = A.double() @ W.double().t() + (b.double() if bias else 0.0)
    assert_close(out, ref, 1e-5, 2e-5, "small-problem product")
    assert torch.isnan(out_w[:, :2]).all() and torch.isnan(out_w[:, 2 + N:]).all(), "wrote outside its column block"
    out2 = torch.empty(M, N, device=dev)
    hip.gemm_ex(A, Bop, out2, bias=b, b_kmajor=km)
    assert torch.equal(out2, out.contiguous()), "two runs differ"


@pytest.mark.parametrize("act", [None, "tanh", "silu", "elu", "ssp", "relu"])
def test_gemm_ex_two_segments_epilogues(dev, act):
    """Two K segments written into a column block of a wider buffer (ldc > N), row mask, alpha, accumulate; then the
    matching input-gradient product through the activation (EPI_DACT) against autograd."""
    from moleculesde_amd import hip
    g = torch.Generator().manual_seed(3)
    M, K1, K2, N = 1000, 300, 119, 300
    A1 = torch.randn(M, K1, generator=g).to(dev)
    A2w = torch.randn(M, 120, generator=g).to(dev)          # K2 = 119 inside rows of stride 120
    A2 = A2w[:, :K2]
    W1 = (torch.randn(N, K1, generator=g) / 17).to(dev)
    W2 = (torch.randn(N, K2, generator=g) / 11).to(dev)      # rows of 119 floats: unaligned -> scalar path
    b = torch.randn(N, generator=g).to(dev)
    mask = (torch.rand(M, generator=g) > 0.2).float().to(dev)
    wide = torch.zeros(M, 364, device=dev)
    base = torch.randn(M, N, generator=g).to(dev)
    wide[:, 32:32 + N] = base
    hip.gemm_ex(A1, W1, wide[:, 32:32 + N], bias=b, A2=A2, B2=W2, act=act, rowscale=mask, alpha=0.5, accumulate=True)
    z = A1.double() @ W1.double().t() + A2.double() @ W2.double().t() + b.double()
    ref = base.double() + 0.5 * mask.double()[:, None] * _act_ref(act)(z)
    assert_close(wide[:, 32:32 + N], ref, 1e-5, 3e-5, f"two-segment {act}")
    assert float(wide[:, :32].abs().max()) == 0 and float(wide[:, 32 + N:].abs().max()) == 0
    # input gradient through the activation: gA1 = (gY * act'(.)) W1 == (gY W1) only when applied BEFORE the product,
    # so the library form is  gZ = (gOut . Wnext) * act'(saved)  for the layer whose OUTPUT went through act:
    # y = act(x Wa^T); o = y Wb^T  =>  g_pre = (gO Wb) * act'(pre)
    Wa = (torch.randn(64, K1, generator=g) / 17).to(dev)
    Wb = (torch.randn(40, 64, generator=g) / 8).to(dev)
    x = A1.clone().requires_grad_(True)
    pre = x @ Wa.t()
    y = _act_ref(act)(pre)
    o = y @ Wb.t()
    gO = torch.randn(M, 40, generator=g).to(dev)
    (gpre_ref,) = torch.autograd.grad(o, pre, gO)
    saved = pre.detach() if act in ("silu", "ssp") else y.detach()
    gpre = torch.empty(M, 64, device=dev)
    hip.gemm_ex(gO, Wb, gpre, b_kmajor=True, act=act, dact_from=saved if act else None)
    assert_close(gpre, gpre_ref, 1e-4, 1e-5, f"dact {act}")


def test_gemm_ex_groups(dev):
    """Block-diagonal product: 16 groups of [M,32] x [32,32] reading column blocks of one wide input and writing column
    blocks of one wide output (EdgeLayer func_q / func_k second layers of all channels in one launch)."""
    from moleculesde_amd import hip
    g = torch.Generator().manual_seed(5)
    M, G = 3588, 16
    H = torch.randn(M, G * 32, generator=g).to(dev)
    W = (torch.randn(G, 32, 32, generator=g) / 6).to(dev)
    b = torch.randn(G, 32, generator=g).to(dev)
    out = torch.empty(M, G * 32, device=dev)
    hip.gemm_ex(H, W, out, bias=b, groups=G, group_strides=dict(a=32, b=32 * 32, bias=32, c=32), N=32, K=32)
    ref = torch.einsum("mgk,gnk->mgn", H.view(M, G, 32).double(), W.double()) + b.double()
    assert_close(out, ref.reshape(M, G * 32), 1e-5, 2e-5, "grouped")
    # k-major groups: NodeNetwork_dense weights [C, in, out]
    Wv = (torch.randn(8, 16, 16, generator=g) / 4).to(dev)
    x = torch.randn(M, 16, generator=g).to(dev)
    xv = torch.empty(M, 8 * 16, device=dev)
    hip.gemm_ex(x, Wv, xv, b_kmajor=True, groups=8, group_strides=dict(b=256, c=16), N=16, K=16)
    ref = torch.einsum("mk,gkn->mgn", x.double(), Wv.double()).reshape(M, 128)
    assert_close(xv, ref, 1e-5, 2e-5, "grouped k-major")


@pytest.mark.parametrize("act", [None, "tanh", "silu", "elu", "ssp", "relu"])
@pytest.mark.parametrize("km", [False, True])
def test_gemm_ex_skinny_products_every_epilogue(dev, act, km):
    """Tall products with N * K <= 48 K (the dense head's node-level products, edge_network_dense.py:33-82: 3 588 rows against
    16 .. 512 columns) run on gemm_small_kernel with the tiled kernel's whole epilogue: bias + second bias, activation on a
    column range with the pre-activation stored, the derivative of a saved activation, row mask, alpha, accumulation into a
    column block of a wider buffer; grouped (block-diagonal) with the derivative -- each against fp64 torch."""
    from moleculesde_amd import hip
    g = torch.Generator().manual_seed(11)
    M, K, N = 3588, 364, 96
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    Bop = W.t().contiguous() if km else W
    b1, b2 = torch.randn(N, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
    mask = (torch.rand(M, generator=g) > 0.2).float().to(dev)
    # forward form: act on columns 16..80, pre-activation stored, written into a column block of a wider buffer
    wide = torch.full((M, N + 8), float("nan"), device=dev)
    Z = torch.full((M, N), float("nan"), device=dev)
    hip.gemm_ex(A, Bop, wide[:, 4:4 + N], bias=b1, bias2=b2, act=act, act_cols=(16, 80), Z=Z, b_kmajor=km)
    z = A.double() @ W.double().t() + b1.double() + b2.double()
    ref = z.clone()
    ref[:, 16:80] = _act_ref(act)(z[:, 16:80])
    assert_close(Z, z, 1e-5, 2e-5, f"skinny {act}: pre-activation")
    assert_close(wide[:, 4:4 + N], ref, 1e-5, 3e-5, f"skinny {act}: output")
    assert torch.isnan(wide[:, :4]).all() and torch.isnan(wide[:, 4 + N:]).all(), "wrote outside its column block"
    # backward form: derivative of the saved tensor (output for tanh / elu / relu, pre-activation for silu / ssp), row mask,
    # alpha, accumulation
    saved = (_act_ref(act)(z) if act in ("tanh", "elu", "relu") else z).float().contiguous()
    gy = torch.randn(M, 32, generator=g).to(dev)
    Wn = (torch.randn(32, N, generator=g) / 6).to(dev)          # next layer's weight [out, in]: gZ = (gy Wn) * act'(saved)
    base = torch.randn(M, N, generator=g).to(dev)
    acc = base.clone()
    hip.gemm_ex(gy, Wn if km else Wn.t().contiguous(), acc, b_kmajor=km, act=act, dact_from=saved, rowscale=mask, alpha=0.5,
                accumulate=True)
    zz = z.clone().requires_grad_(True)
    _act_ref(act)(zz).backward(gy.double() @ Wn.double())
    assert_close(acc, base.double() + 0.5 * mask.double()[:, None] * zz.grad, 1e-5, 3e-5, f"skinny {act}: derivative form")
    # grouped: 16 groups of [M, 32] x [32, 32] with the derivative of a saved tensor (func_q / func_k backward)
    G = 16
    gq = torch.randn(M, G * 32, generator=g).to(dev)
    Wg = (torch.randn(G, 32, 32, generator=g) / 6).to(dev)
    Hs = torch.tanh(torch.randn(M, G * 32, generator=g)).to(dev) if act in (None, "tanh") else torch.randn(M, G * 32, generator=g).to(dev)
    out = torch.full((M, G * 32), float("nan"), device=dev)
    hip.gemm_ex(gq, Wg if km else Wg.transpose(1, 2).contiguous(), out, b_kmajor=km, groups=G,
                group_strides=dict(a=32, b=1024, c=32, r=32), N=32, K=32, act=act, dact_from=Hs)
    lin = torch.einsum("mgn,gnk->mgk", gq.view(M, G, 32).double(), Wg.double()).reshape(M, G * 32)
    dd_ = {None: lambda r: torch.ones_like(r), "tanh": lambda r: 1 - r * r, "relu": lambda r: (r > 0).double(),
           "elu": lambda r: torch.where(r > 0, torch.ones_like(r), r + 1), "ssp": torch.sigmoid,
           "silu": lambda r: torch.sigmoid(r) * (1 + r * (1 - torch.sigmoid(r)))}[act]
    assert_close(out, lin * dd_(Hs.double()), 1e-5, 3e-5, f"skinny grouped {act}")


# ---- row-strip GEMM family (csrc/gemm_rs.hip) ----------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(3588, 300, 300), (3588, 600, 300), (3588, 300, 600), (3588, 128, 300), (3588, 300, 128),
                                   (3588, 32, 300), (3588, 128, 32), (35186, 32, 300), (35186, 128, 32), (35186, 32, 128),
                                   (35186, 300, 32), (9000, 128, 64), (100, 20, 36), (17, 300, 300), (1, 16, 4),
                                   (3588, 728, 364), (4096, 364, 120), (777, 80, 64), (500, 176, 32), (3588, 256, 128)])
def test_gemm_rs_plain(dev, M, N, K):
    """C = act(A B + bias) with the pre-activation stored, against fp64 torch; [K][N] weights, strips with row tails,
    16-column tile tails and interleaved column segments (N = 300, 364, 20), K tails (K = 300, 36, 4), every
    tiles-per-wave variant.  NaN-filled outputs prove every element is written."""
    from moleculesde_amd import hip
    km = True
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    Bop = W.t().contiguous() if km else W
    out = torch.full((M, N), float("nan"), device=dev)
    Z = torch.full((M, N), float("nan"), device=dev)
    hip.gemm_rs(A, Bop, out, bias=b, act="ssp", Z=Z, b_kmajor=km, fallback=False)
    zr = A.double() @ W.double().t() + b.double()
    assert_close(Z, zr, 1e-5, 2e-5, "gemm_rs pre-activation")
    assert_close(out, _act_ref("ssp")(zr), 1e-5, 2e-5, "gemm_rs out")
    out2 = torch.full((M, N), float("nan"), device=dev)
    hip.gemm_rs(A, Bop, out2, bias=b, act="ssp", b_kmajor=km, fallback=False)
    assert torch.equal(out, out2), "gemm_rs must be bitwise reproducible"


@pytest.mark.parametrize("M,N,K", [(3588, 300, 300), (3588, 600, 300), (3588, 300, 600), (3588, 128, 300), (3588, 300, 128),
                                   (3588, 32, 300), (3588, 128, 32), (3588, 728, 364), (3588, 728, 728), (35186, 32, 300),
                                   (35186, 300, 32), (777, 80, 64), (512, 176, 36), (9000, 128, 64), (3588, 256, 100),
                                   (1030, 36, 4), (3588, 304, 16)])
def test_gemm_t2_plain(dev, M, N, K):
    """msde_gemm_t2 (2-D tiles, both operands through LDS by LDS-DMA): C = act(A W^T + bias) + res with the pre-activation
    stored, against fp64 torch; [N][K] weights, row / column / K tails (K = 300: a 12-wide last K tile whose second k-half
    is skipped; K = 36, 100, 4; odd and even tile counts), every tiles-per-workgroup variant, forced split counts.
    NaN-filled outputs prove every element is written; two launches must agree bit for bit."""
    from moleculesde_amd import hip, _lib
    if not _lib.load().msde_gemm_t2_supported(M, N, K, 0):
        pytest.skip("shape not taken by msde_gemm_t2")
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    zr = A.double() @ W.double().t() + b.double()
    ref = _act_ref("ssp")(zr) + res.double()
    for splits in (0, 1, 3, 8):
        ntiles = (N + 15) // 16
        if splits > ntiles:
            continue
        out = torch.full((M, N), float("nan"), device=dev)
        Z = torch.full((M, N), float("nan"), device=dev)
        hip.gemm_rs(A, W, out, bias=b, act="ssp", Z=Z, res=res, t2=True, splits=splits)
        assert_close(Z, zr, 1e-5, 2e-5, f"gemm_t2 pre-activation (splits {splits})")
        assert_close(out, ref, 1e-5, 2e-5, f"gemm_t2 out (splits {splits})")
        out2 = torch.full((M, N), float("nan"), device=dev)
        hip.gemm_rs(A, W, out2, bias=b, act="ssp", res=res, t2=True, splits=splits)
        assert torch.equal(out, out2), "gemm_t2 must be bitwise reproducible"


@pytest.mark.parametrize("act", [None, "silu", "ssp", "relu", "tanh"])
def test_gemm_rs_epilogues(dev, act):
    """Residual + accumulate into a column block of a wider buffer, and the input-gradient product through an activation
    (EPI_DACT) with the residual gradient added -- against autograd."""
    from moleculesde_amd import hip
    g = torch.Generator().manual_seed(11)
    M, K, N = 1000, 300, 300
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / 17).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    wide = torch.zeros(M, 364, device=dev)
    base = torch.randn(M, N, generator=g).to(dev)
    wide[:, 32:32 + N] = base
    hip.gemm_rs(A, W.t().contiguous(), wide[:, 32:32 + N], bias=b, act=act, res=res, accumulate=True, b_kmajor=True,
                fallback=False)
    ref = base.double() + res.double() + _act_ref(act)(A.double() @ W.double().t() + b.double())
    assert_close(wide[:, 32:32 + N], ref, 1e-5, 3e-5, f"rs residual+accumulate {act}")
    assert float(wide[:, :32].abs().max()) == 0 and float(wide[:, 32 + N:].abs().max()) == 0
    Wa = (torch.randn(64, K, generator=g) / 17).to(dev)
    Wb = (torch.randn(40, 64, generator=g) / 8).to(dev)
    x = A.clone().requires_grad_(True)
    pre = x @ Wa.t()
    y = _act_ref(act)(pre)
    o = y @ Wb.t()
    gO = torch.randn(M, 40, generator=g).to(dev)
    (gpre_ref,) = torch.autograd.grad(o, pre, gO)
    saved = pre.detach() if act in ("silu", "ssp") else y.detach()
    r2 = torch.randn(M, 64, generator=g).to(dev)
    gpre = torch.empty(M, 64, device=dev)
    hip.gemm_rs(gO, Wb, gpre, b_kmajor=True, act=act, dact_from=saved if act else None, res=r2, fallback=False)
    assert_close(gpre, gpre_ref.double() + r2.double(), 1e-4, 1e-5, f"rs dact {act}")


@pytest.mark.parametrize("t2", [False, True], ids=["strips", "tiles"])
@pytest.mark.parametrize("M,C1,C2,relu,bound", [(3588, 600, 300, True, None), (3588, 300, 600, False, None),
                                                (35186, 300, 32, True, None), (200, 64, 48, True, 150),
                                                (3648, 600, 300, True, 3588), (1350, 600, 300, True, None),
                                                (1408, 300, 600, True, 1350), (700, 128, 64, False, None)])
def test_gemm_rs_fused_batchnorm_chain(dev, M, C1, C2, relu, bound, t2):
    """x -> Linear(K0, C1) -> BatchNorm1d (training) -> [ReLU] -> Linear(C1, C2) with the statistics in the first product's
    epilogue, msde_bn_fin_fwd, and the BatchNorm apply (+ ReLU) in the second product's A load; then the backward:
    input-gradient product with the ReLU gate and the BatchNorm-backward partial sums in its epilogue, msde_bn_fin_bwd,
    and the BatchNorm input gradient formed in the A load of the next input-gradient product.  Against torch CPU
    autograd in fp64 of the same fp32 inputs.  `bound`: only the first `bound` rows are valid (capacity buckets)."""
    from moleculesde_amd import hip, _lib
    import ctypes
    K0 = 128
    if t2 and M < 512:
        pytest.skip("msde_gemm_t2 takes M >= 512")

    def prod(A, W, out, forward, **kw):          # W: nn.Linear layout [out][in]; both kernels, each with the layout it reads
        if t2:
            return hip.gemm_rs(A, W if forward else W.t().contiguous(), out, t2=True, **kw)
        return hip.gemm_rs(A, W.t().contiguous() if forward else W, out, b_kmajor=True, fallback=False, **kw)

    def geometry(M_, N_, K_):
        a_, b_ = ctypes.c_int(0), ctypes.c_int(0)
        _lib.call("msde_gemm_t2_geometry" if t2 else "msde_gemm_rs_geometry", M_, N_, K_, ctypes.byref(a_), ctypes.byref(b_))
        return a_.value, b_.value
    torch.manual_seed(M + C1)
    Mv = bound or M
    x = torch.randn(M, K0)
    W1 = torch.randn(C1, K0) / K0 ** 0.5
    b1 = torch.randn(C1) * 0.1
    gamma, beta = torch.randn(C1) * 0.3 + 1, torch.randn(C1) * 0.3
    W2 = torch.randn(C2, C1) / C1 ** 0.5
    gy = torch.randn(M, C2)
    gy[Mv:] = 0
    # reference (valid rows only)
    xr = x[:Mv].double().requires_grad_(True)
    W1r, b1r, gr, br, W2r = (t.double().requires_grad_(True) for t in (W1, b1, gamma, beta, W2))
    z = xr @ W1r.t() + b1r
    mu, var = z.mean(0), z.var(0, unbiased=False)
    xhat = (z - mu) / torch.sqrt(var + 1e-5)
    a = xhat * gr + br
    if relu:
        a = torch.relu(a)
    y = a @ W2r.t()
    y.backward(gy[:Mv].double())

    xd, W1d, b1d, gd, bd, W2d, gyd = (t.to(dev) for t in (x, W1, b1, gamma, beta, W2, gy))
    mvd = torch.tensor([Mv], dtype=torch.int32, device=dev) if bound else None
    st = hip._stream()
    p = hip._p
    # forward
    strips, srows = geometry(M, C1, K0)
    stats = torch.full((strips, 2, C1), float("nan"), device=dev)
    zd = torch.empty(M, C1, device=dev)
    prod(xd, W1d, zd, True, bias=b1d, stats=stats, stats_mode="bnfwd", m_valid=mvd)
    scale, shift, smean, srstd = (torch.empty(C1, device=dev) for _ in range(4))
    rm, rv = torch.zeros(C1, device=dev), torch.ones(C1, device=dev)
    _lib.call("msde_bn_fin_fwd", p(stats), strips, srows, M, p(mvd), C1, p(gd), p(bd), 1e-5, 0.1, p(rm), p(rv), p(scale),
              p(shift), p(smean), p(srstd), st)
    ad = torch.full((M, C1), float("nan"), device=dev)
    yd = torch.empty(M, C2, device=dev)
    prod(zd, W2d, yd, True, axf="affine", xf=(scale, shift), relu=relu, A_out=ad, m_valid=mvd)
    yd2, ad2 = torch.full_like(yd, float("nan")), torch.full_like(ad, float("nan"))
    prod(zd, W2d, yd2, True, axf="affine", xf=(scale, shift), relu=relu, A_out=ad2, m_valid=mvd)
    assert torch.equal(yd, yd2) and torch.equal(ad, ad2), "fused BatchNorm-apply product must be bitwise reproducible"
    assert_close(smean, mu.detach(), 1e-5, 1e-5, "fused bn mean")
    assert_close(srstd, 1 / torch.sqrt(var.detach() + 1e-5), 1e-4, 1e-5, "fused bn rstd")
    assert_close(ad[:Mv], a.detach(), 1e-4, 2e-5, "fused bn apply (A_out)")
    assert_close(yd[:Mv], y.detach(), 1e-4, 5e-5, "fused bn chain out")
    unb = var.detach() * Mv / max(Mv - 1, 1)
    assert_close(rm, 0.1 * mu.detach(), 1e-5, 1e-6, "running mean")
    assert_close(rv, 0.9 + 0.1 * unb, 1e-4, 1e-6, "running var")
    # backward: g_a = gy W2 gated by the ReLU, with the BatchNorm-backward partial sums
    strips2, _ = geometry(M, C1, C2)
    stats2 = torch.full((strips2, 2, C1), float("nan"), device=dev)
    gad = torch.empty(M, C1, device=dev)
    prod(gyd, W2d, gad, False, act="relu" if relu else None, dact_from=ad if relu else None, stats=stats2,
         stats_mode="bnbwd", stats_z=zd, stats_mean=smean, m_valid=mvd)
    pv, wv, uv, dgam, dbet = (torch.empty(C1, device=dev) for _ in range(5))
    _lib.call("msde_bn_fin_bwd", p(stats2), strips2, M, p(mvd), C1, p(gd), p(smean), p(srstd), p(pv), p(wv), p(uv), p(dgam),
              p(dbet), st)
    dzd = torch.full((M, C1), float("nan"), device=dev)
    gxd = torch.empty(M, K0, device=dev)
    prod(gad, W1d, gxd, False, axf="bnbwd", xf=(pv, wv, uv), A2=zd, A_out=dzd, m_valid=mvd)
    gxd2, dzd2 = torch.full_like(gxd, float("nan")), torch.full_like(dzd, float("nan"))
    prod(gad, W1d, gxd2, False, axf="bnbwd", xf=(pv, wv, uv), A2=zd, A_out=dzd2, m_valid=mvd)
    assert torch.equal(gxd, gxd2) and torch.equal(dzd, dzd2), "fused BatchNorm-backward product must be bitwise reproducible"
    # a pre-activation within rounding of the ReLU gate may fall on either side: compare away from it
    apre = (xhat * gr + br).detach()
    unsure = (apre.abs() <= 2e-5) if relu else torch.zeros_like(apre, dtype=torch.bool)
    assert int(unsure.sum()) <= 1e-4 * unsure.numel() + 2
    # an element on the gate boundary contributes its whole upstream gradient or nothing: allow for it column by column
    ga_ref = (gy[:Mv].double() @ W2.double())
    allow_b = (unsure * ga_ref.abs()).sum(0)
    allow_g = (unsure * (ga_ref * xhat.detach()).abs()).sum(0)
    for got, ref, allow, what in ((dgam, gr.grad, allow_g, "fused dgamma"), (dbet, br.grad, allow_b, "fused dbeta")):
        err = (got.double().cpu() - ref).abs()
        bnd = 2e-4 * ref.abs() + 2e-5 * float(ref.abs().max()) + 1e-6 + 1.01 * allow
        assert bool((err <= bnd).all()), f"{what}: max excess {(err - bnd).max().item():.3e}"
    if not bool(unsure.any()):
        assert_close(gxd[:Mv], xr.grad, 2e-3, 2e-5 * float(xr.grad.abs().max()) + 1e-6, "fused bn chain dx")
        gW1 = dzd[:Mv].double().cpu().t() @ x[:Mv].double()
        assert_close(gW1, W1r.grad, 2e-3, 2e-5 * float(W1r.grad.abs().max()) + 1e-6, "dz (A_out) -> gW1")


@pytest.mark.parametrize("B,n3", [(1, 63), (4, 252), (37, 5000)])
def test_l1_energy_force_loss(dev, B, n3):
    """msde_l1_energy_force_loss (finetune_MD17.py:68-74: both L1 losses and the seeds of the backward pass in one launch)
    against torch autograd on the same values, including exact zeros of the residual (sign(0) = 0)."""
    from moleculesde_amd import _lib, hip
    g = torch.Generator().manual_seed(B + n3)
    E, y = torch.randn(B, generator=g).to(dev), torch.randn(B, generator=g).to(dev)
    dE, f = torch.randn(n3, generator=g).to(dev), torch.randn(n3, generator=g).to(dev)
    y[0] = E[0]
    f[1] = -dE[1]
    Er, dr = E.clone().requires_grad_(True), dE.clone().requires_grad_(True)
    ref = 0.05 * (Er - y).abs().mean() + 0.95 * ((-dr) - f).abs().mean()
    ge_ref, gd_ref = torch.autograd.grad(ref, [Er, dr])
    loss = torch.empty(1, device=dev)
    gE, gd = torch.empty_like(E), torch.empty_like(dE)
    _lib.call("msde_l1_energy_force_loss", hip._p(E), hip._p(y), B, hip._p(dE), hip._p(f), n3, -1.0, 0.05, 0.95, hip._p(loss),
              hip._p(gE), hip._p(gd), hip._stream())
    assert_close(loss[0], ref.detach(), 1e-6, 1e-7, "loss")
    assert_close(gE, ge_ref, 1e-6, 1e-9, "d loss / d energy")
    assert_close(gd, gd_ref, 1e-6, 1e-9, "d loss / d (dE)")
    assert float(gE[0]) == 0.0 and float(gd[1]) == 0.0


@pytest.mark.parametrize("M,C,strips,gate", [(3588, 600, 57, False), (3588, 300, 57, True), (3712, 300, 228, True), (100, 64, 2, False),
                                             (1000, 20, 16, True)])
def test_bn_bwd_fin_cols_matches_finish_plus_pass(dev, M, C, strips, gate):
    """msde_bn_bwd_fin_cols (the BatchNorm-backward finish inside the column pass: one launch on the GIN backward chain,
    molecule_gnn_model.py:17,176-182) against msde_bn_fin_bwd + msde_bn_bwd_cols: the input gradient, dgamma and dbeta -- same
    formulas, another (fixed) summation order over the strips; rows behind the row bound are written as zero; two runs bit-equal."""
    from moleculesde_amd import hip, _lib
    g = torch.Generator().manual_seed(M + C + strips)
    stats = torch.randn(strips, 2, C, generator=g).to(dev)
    gamma, mean = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    rstd = (torch.rand(C, generator=g) + 0.5).to(dev)
    Gw = torch.randn(M, C + 8, generator=g).to(dev)
    G = Gw[:, 4:4 + C]                                    # a column block of a wider buffer (row stride != C)
    Z = torch.randn(M, C, generator=g).to(dev)
    xf3 = torch.randn(C, generator=g).to(dev) if gate else None
    xf4 = torch.randn(C, generator=g).to(dev) if gate else None
    mv = M - 37
    with hip.row_bounds({M: torch.tensor([mv], dtype=torch.int32, device=dev)}):
        rows = hip.bound_tensor(M)
        pw, gb = hip._bn_fin_bwd(stats, strips, M, C, gamma, mean, rstd)
        ref = torch.full((M, C), float("nan"), device=dev)
        _lib.call("msde_bn_bwd_cols", hip._p(G), hip._ld(G), hip._p(Z), hip._ld(Z), hip._p(pw[0]), hip._p(pw[1]), hip._p(pw[2]),
                  hip._p(xf3), hip._p(xf4), M, hip._p(rows), C, hip._p(ref), C, hip._stream())
        out = torch.full((M, C), float("nan"), device=dev)
        gb2 = hip._bn_bwd_fin_cols(stats, strips, M, C, gamma, mean, rstd, G, Z, xf3, xf4, out, rows)
        out_b = torch.empty(M, C, device=dev)
        gb3 = hip._bn_bwd_fin_cols(stats, strips, M, C, gamma, mean, rstd, G, Z, xf3, xf4, out_b, rows)
    assert torch.isfinite(out).all() and float(out[mv:].abs().max()) == 0.0
    assert_close(out, ref.double(), 2e-5, 2e-5, "fused finish + pass: input gradient")
    assert_close(gb2, gb.double(), 2e-5, 2e-5, "fused finish + pass: dgamma / dbeta")
    assert torch.equal(out, out_b) and torch.equal(gb2, gb3), "two runs differ"


def test_fanout_sums_column_blocks_in_one_launch(dev):
    """dd.fanout: a tensor with n consumers gets ONE gradient kernel; consumers' gradients that are column blocks of wider
    buffers (the edge half of the basis MLP's input gradient, equivariant_scorenetwork.py:154-157) are summed where they lie
    (msde_dd_sum_rows_n).  Same additions in the same order as autograd's accumulation: bit-exact."""
    from moleculesde_amd import dd
    torch.manual_seed(21)
    E, D = 1237, 32
    x = torch.randn(E, D)
    wide1, wide2, plain = torch.randn(E, 2 * D), torch.randn(E, D + 8), torch.randn(E, D)
    ref = (wide1[:, D:] + wide2[:, 8:]) + plain                  # index order of the fan-out node
    xd = x.to(dev).requires_grad_(True)
    a, b, c = dd.fanout(xd, 3)
    # consumers whose backward hands on views of wider buffers / a contiguous tensor
    w1, w2, pl = wide1.to(dev), wide2.to(dev), plain.to(dev)
    torch.autograd.backward([a, b, c], [w1[:, D:], w2[:, 8:], pl])
    assert_close(xd.grad, ref, 0, 0, "fanout strided gradients")
    # all contiguous: the flat kernel
    xd2 = x.to(dev).requires_grad_(True)
    a, b = dd.fanout(xd2, 2)
    torch.autograd.backward([a, b], [pl, w1[:, :D].contiguous()])
    assert_close(xd2.grad, plain + wide1[:, :D], 0, 0, "fanout contiguous gradients")
    # a consumer that does not contribute
    xd3 = x.to(dev).requires_grad_(True)
    a, b, c = dd.fanout(xd3, 3)
    (a * 2.0 + c).sum().backward()
    assert_close(xd3.grad, torch.full((E, D), 3.0), 0, 0, "fanout with an unused alias")
    # the C entry point refuses operands it cannot vectorise
    import ctypes
    from moleculesde_amd import _lib
    lib = _lib.load()
    arr = (ctypes.c_void_p * 2)(w1.data_ptr() + 4, pl.data_ptr())
    lds = (ctypes.c_int * 2)(2 * D, D)
    y = torch.empty(E, D, device=dev)
    assert lib.msde_dd_sum_rows_n(ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(lds, ctypes.c_void_p), 2, E, D,
                                  ctypes.c_void_p(y.data_ptr()), None) != 0


def test_grouped_wgrad_narrow_tile_shapes(dev):
    """Weight / bias gradients of layers with a 32-wide (or narrower) side run on their own tile shapes inside the grouped launch
    (32 x 128 and 128 x 32: the four waves side by side; 32 x 32 with the rows of a 128-row K tile split over the waves --
    msde_linear_bwd_w_describe_ld): each against the fp64 product, row counts that are not multiples of the K tile, widths of 8 and
    16 (columns beyond the operand read from its last float4, never stored), and bit-identical on a second run."""
    from moleculesde_amd import hip
    torch.manual_seed(33)
    shapes = [(35186, 32, 300), (5000, 32, 128), (35186, 128, 32), (4099, 32, 72), (3588, 16, 364), (3588, 16, 16),
              (52680, 16, 32), (1000, 32, 32), (333, 32, 32), (777, 8, 200), (2049, 300, 8), (3588, 32, 64)]
    layers = []
    for M, N, K in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev, requires_grad=True)
        b = torch.randn(N, device=dev, requires_grad=True)
        g = torch.randn(M, N, device=dev)
        layers.append((x, w, b, g))

    def run():
        for x, w, b, g in layers:
            w.grad = b.grad = None
        slabs.begin_param_grad_batch()
        for x, w, b, g in layers:
            hip.linear(x, w, b).backward(g)
        slabs.finish_param_grad_batch()
        torch.cuda.synchronize()
        return [(w.grad.clone(), b.grad.clone()) for x, w, b, g in layers]

    got, again = run(), run()
    for (M, N, K), (x, w, b, g), (gw, gb), (gw2, gb2) in zip(shapes, layers, got, again):
        ref = g.double().t() @ x.double()
        refb = g.double().sum(0)
        ew = float((gw.double() - ref).abs().max() / ref.abs().max())
        eb = float((gb.double() - refb).abs().max() / refb.abs().max())
        assert ew < 5e-6 and eb < 5e-6, (M, N, K, ew, eb)          # fp32 accumulation over <= 52 680 rows against fp64
        assert torch.equal(gw, gw2) and torch.equal(gb, gb2), (M, N, K)
