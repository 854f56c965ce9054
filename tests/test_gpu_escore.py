"""EquivariantScoreNetwork on the one-workgroup-per-molecule kernels (csrc/escore_mol.hip, moleculesde_amd/escore.py)
against (a) the oracle's CPU restatement of equivariant_scorenetwork.py:121-169 and (b) the operator-by-operator HIP path,
which draws the identical dropout masks from the same seeds.  fp32; tolerances at each assert."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import restate as R  # noqa: E402
from helpers import assert_close  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from moleculesde_amd import _lib
    _lib.load()
    return torch.device("cuda", 0)


def _case(dev, B, seed):
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import plan as P
    from moleculesde_amd.geom3d import sde_2d_to_3d as M
    from moleculesde_amd.synthetic import make_batch
    torch.manual_seed(seed)
    cpu_b = make_batch(B, seed)
    dev_b = G.prepare_batch(cpu_b.clone(), dev)
    pl = P.get_plan(dev_b)
    ep = pl.ext
    net = M.EquivariantScoreNetwork(32, hidden_coff_dim=128)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.normal_(1, 0.3)
                m.bias.normal_(0, 0.3)
    net = net.to(dev)
    x = torch.randn(ep.N, 32)
    ea_canon = torch.randn(ep.E, 32) * 0.7
    basis = torch.randn(ep.E, 9)
    return cpu_b, pl, ep, net, x, ea_canon, basis


def _oracle(net, cpu_b, ep, x, ea_canon, basis):
    o = R.EquivariantScoreNetwork(32, hidden_coff_dim=128)
    o.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    o.eval()
    perm = ep.perm_t.cpu()                              # canonical edge id -> position in the caller's edge_index
    ei = cpu_b.extended_edge_index
    ea = torch.empty_like(ea_canon)
    bs = torch.empty_like(basis)
    ea[perm] = ea_canon
    bs[perm] = basis
    return o, ei, ea, [bs[:, 0:3], bs[:, 3:6], bs[:, 6:9]]


@pytest.mark.parametrize("B,seed", [(1, 3), (7, 5), (256, 0)])
def test_escore_mol_forward_eval_vs_oracle_and_operator_path(dev, B, seed):
    from moleculesde_amd.geom3d import sde_2d_to_3d as M
    cpu_b, pl, ep, net, x, ea, basis = _case(dev, B, seed)
    net.eval()
    o, ei, ea_o, basis_o = _oracle(net, cpu_b, ep, x, ea, basis)
    with torch.no_grad():
        ref = o(ei, x, ea_o, basis_o)["gradient"]
        xd, ed, bd = x.to(dev), ea.to(dev), basis.to(dev)
        assert M.MOL_KERNEL
        got = net(ep, xd, ed, bd, pl)["gradient"]
        M.MOL_KERNEL = False
        try:
            ops = net(ep, xd, ed, bd, pl)["gradient"]
        finally:
            M.MOL_KERNEL = True
    scale = float(ref.abs().max())
    assert_close(got, ref, 1e-4, 2e-5 * scale, "escore mol kernel vs oracle")
    assert_close(got, ops, 1e-5, 2e-6 * scale, "escore mol kernel vs operator path")
    # bit-reproducible: every sum inside the kernel has a fixed order
    with torch.no_grad():
        again = net(ep, xd, ed, bd, pl)["gradient"]
    assert torch.equal(got, again)


def test_escore_mol_forward_dropout_masks_match_operator_path(dev):
    from moleculesde_amd.geom3d import sde_2d_to_3d as M
    cpu_b, pl, ep, net, x, ea, basis = _case(dev, 16, 11)
    net.train()
    xd, ed, bd = x.to(dev), ea.to(dev), basis.to(dev)
    ctr = torch.full((1,), 9, dtype=torch.int64, device=dev)
    outs = []
    for seed_dev in (None, ctr):
        net.seed_dev = seed_dev
        pair = []
        for mol in (True, False):
            net._calls = 41
            M.MOL_KERNEL = mol
            try:
                with torch.no_grad():
                    pair.append(net(ep, xd, ed, bd, pl)["gradient"])
            finally:
                M.MOL_KERNEL = True
        assert_close(pair[0], pair[1], 1e-5, 2e-6 * float(pair[1].abs().max()), "escore dropout: mol kernel vs operator path")
        outs.append(pair[0])
    net.seed_dev = None
    assert not torch.allclose(outs[0], outs[1])         # the device counter changes the masks
    net.eval()
    with torch.no_grad():
        clean = net(ep, xd, ed, bd, pl)["gradient"]
    assert not torch.allclose(outs[0], clean)           # and dropout was really on


def test_escore_mol_padded_rows_and_empty_molecules(dev):
    """Capacity padding: atoms behind the last molecule get zero scores, an empty molecule in the middle is skipped."""
    from moleculesde_amd import escore
    cpu_b, pl, ep, net, x, ea, basis = _case(dev, 5, 21)
    net.eval()
    xd, ed, bd = x.to(dev), ea.to(dev), basis.to(dev)
    with torch.no_grad():
        ref = escore.forward_nograd(net, ep, pl, xd, ed, bd, 0, None)
    import types
    N, pad = ep.N, 7
    mp = pl.mol_ptr.cpu().tolist()
    mp2 = mp[:3] + [mp[2]] + mp[3:]                     # an empty molecule between molecules 1 and 2
    pl2 = types.SimpleNamespace(mol_ptr=torch.tensor(mp2, dtype=torch.int32, device=dev), B=len(mp2) - 1, N_max=pl.N_max)
    ep2 = types.SimpleNamespace(N=N + pad, E=ep.E, src=ep.src, dst=ep.dst,
                                rowptr=torch.cat([ep.rowptr, ep.rowptr[-1:].expand(pad)]).contiguous())
    x2 = torch.cat([xd, torch.full((pad, 32), float("nan"), device=dev)])
    with torch.no_grad():
        got = escore.forward_nograd(net, ep2, pl2, x2, ed, bd, 0, None)
    assert torch.equal(got[:N], ref)
    assert torch.equal(got[N:], torch.zeros(pad, 3, device=dev))


def _grads(net, ep, pl, xd, ed, bd, w, mol):
    from moleculesde_amd.geom3d import sde_2d_to_3d as M
    net.zero_grad(set_to_none=True)
    x = xd.clone().requires_grad_(True)
    e = ed.clone().requires_grad_(True)
    M.MOL_KERNEL_TRAIN = mol
    try:
        out = net(ep, x, e, bd, pl)["gradient"]
        (out * w).sum().backward()
    finally:
        M.MOL_KERNEL_TRAIN = False
    return out.detach(), x.grad, e.grad, {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("B,seed,train", [(1, 3, False), (7, 5, False), (40, 9, True), (256, 0, True)])
def test_escore_mol_backward_vs_operator_path_and_oracle(dev, B, seed, train):
    """All gradients of the one-launch backward (node features, edge features, every parameter) against the operator path
    (same dropout masks when training) and, without dropout, against the oracle's autograd on the CPU."""
    cpu_b, pl, ep, net, x, ea, basis = _case(dev, B, seed)
    net.train(train)
    xd, ed, bd = x.to(dev), ea.to(dev), basis.to(dev)
    w = torch.randn(ep.N, 3)
    wd = w.to(dev)
    net._calls = 7
    o1, gx1, ge1, gp1 = _grads(net, ep, pl, xd, ed, bd, wd, True)
    net._calls = 7
    o2, gx2, ge2, gp2 = _grads(net, ep, pl, xd, ed, bd, wd, False)
    assert_close(o1, o2, 1e-5, 2e-6 * float(o2.abs().max()), "forward (training variant)")
    assert_close(gx1, gx2, 1e-4, 2e-5 * float(gx2.abs().max()), "g node features")
    assert_close(ge1, ge2, 1e-4, 2e-5 * float(ge2.abs().max()), "g edge features")
    assert set(gp1) == set(gp2)
    scale = max(float(v.abs().max()) for v in gp2.values())
    for k in gp2:
        assert_close(gp1[k], gp2[k], 1e-4, 2e-5 * scale, f"grad {k}")
    if not train:
        o, ei, ea_o, basis_o = _oracle(net, cpu_b, ep, x, ea, basis)
        xo = x.clone().requires_grad_(True)
        eo = ea_o.clone().requires_grad_(True)
        (o(ei, xo, eo, basis_o)["gradient"] * w).sum().backward()
        perm = ep.perm_t.cpu()
        assert_close(gx1, xo.grad, 2e-4, 5e-5 * float(xo.grad.abs().max()), "g node features vs oracle")
        assert_close(ge1, eo.grad[perm], 2e-4, 5e-5 * float(eo.grad.abs().max()), "g edge features vs oracle")
        og = {k: p.grad for k, p in o.named_parameters() if p.grad is not None}
        oscale = max(float(v.abs().max()) for v in og.values())
        for k, v in og.items():
            assert_close(gp1[k], v, 2e-4, 5e-5 * oscale, f"grad {k} vs oracle")
    # bit-reproducible backward
    net._calls = 7
    _, gx3, ge3, gp3 = _grads(net, ep, pl, xd, ed, bd, wd, True)
    assert torch.equal(gx1, gx3) and torch.equal(ge1, ge3) and all(torch.equal(gp1[k], gp3[k]) for k in gp1)


def test_trainer_score_kernel_mol_matches_ops(dev):
    """pretrain.Trainer with --score_kernel mol (score network forward + backward as one launch each) against the default
    operator path: same seeds -> same noise and dropout masks -> same losses and the same parameters after two Adam steps,
    eagerly and through the captured hipGraph."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import pretrain
    from moleculesde_amd.synthetic import make_batch
    cpu = [make_batch(24, seed=s) for s in (3, 4)]
    res = {}
    for kind in ("ops", "mol"):
        torch.manual_seed(5)
        tr = pretrain.Trainer(pretrain.readme_args(emb_dim=64, SDE_coeff_generative_3Dto2D=0, lr=1e-3, batch_size=24,
                                                   score_kernel=kind), dev)
        assert tr.models["SDE_2Dto3D_model"].score_network.mol_kernel_train == (kind == "mol")
        bs = [G.prepare_batch(b.clone(), dev) for b in cpu]
        losses = [float(tr.step(b)[0]) for b in bs]
        tr.capture(bs[0])
        losses.append(float(tr.step_graph(bs[0])))
        torch.cuda.synchronize()
        res[kind] = (losses, tr.opt.flat_p.detach().clone())
    for a, b in zip(res["mol"][0], res["ops"][0]):
        assert abs(a - b) <= 2e-5 * abs(b), (res["mol"][0], res["ops"][0])
    pa, pb = res["mol"][1], res["ops"][1]
    # Adam's normalised update turns the rounding noise of an analytically ZERO gradient (key biases under the softmax) into
    # steps of +-lr, with either sign in either trainer: up to 2 lr per step, three steps at lr = 1e-3; everything else agrees
    # to rounding, which the mean shows
    d = (pa - pb).abs()
    assert float(d.max()) <= 6.5e-3, float(d.max())
    assert float(d.mean()) <= 2e-5, float(d.mean())


@pytest.mark.parametrize("cls,sizes", [("SDEModel2Dto3D_02", [12, 5, 20, 9, 1, 17]), ("SDEModel2Dto3D_01", [14] * 10),
                                        ("SDEModel2Dto3D_02", [20] * 3 + [2]), ("SDEModel2Dto3D_02", [31, 3, 24, 16])])
def test_get_score_fused_matches_operator_path(dev, cls, sizes):
    """msde_escore_mol_score -- two launches (per-edge work: frame, Fourier features, input_mlp / coff_mlp / project, lin_edge of
    the four layers, the edge half of the basis MLPs, in a wide launch; then one workgroup per molecule),
    SDE_model_2D_to_3D.py:393-445 -- against the same get_score on the separate geometry launches +
    msde_escore_mol_fwd and against the plain operator path.  The in-kernel Fourier features use the hardware sin on the phase
    in revolutions (|error| ~1e-6 per feature): 2e-4 relative to the score's scale."""
    import numpy as np
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import escore, plan as P
    from moleculesde_amd.batch import Batch
    from moleculesde_amd.geom3d import sde_2d_to_3d as M
    from moleculesde_amd.synthetic import make_molecule
    torch.manual_seed(11)
    rng = np.random.default_rng(5)
    b = G.prepare_batch(Batch.from_data_list([make_molecule(rng, n) for n in sizes]), dev)
    gnn = G.GNN(3, 32, gnn_type="GIN").to(dev).eval()
    model = getattr(G, cls)(emb_dim=32, hidden_dim=32, beta_min=0.1, beta_max=1.0, num_diffusion_timesteps=1000,
                            beta_schedule=None, SDE_type="VE", use_extend_graph=True).to(dev).eval()
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.normal_(1, 0.3)
                m.bias.normal_(0, 0.3)
        rep = gnn(b.x, b.edge_index, b.edge_attr)
    pl = P.get_plan(b)
    outs = {}
    # coordinates at a small and a large diffusion time
    poses = [(b.positions + scale * torch.randn_like(b.positions)).contiguous() for scale in (0.3, 6.0)]
    keep = (M.MOL_KERNEL, M.MOL_KERNEL_SCORE)
    modes = [("two_launches", True, True), ("mol_fwd", True, False), ("ops", False, False)]
    try:
        for name, mk, ms in modes:
            M.MOL_KERNEL, M.MOL_KERNEL_SCORE = mk, ms
            if ms:
                assert escore.score_supported(model, pl)
            outs[name] = [model.get_score_raw(rep, b, pos).clone() for pos in poses]
    finally:
        M.MOL_KERNEL, M.MOL_KERNEL_SCORE = keep
    for i in range(2):
        ref = outs["ops"][i]
        assert torch.isfinite(ref).all()
        for name in outs:
            if name != "ops":
                assert_close(outs[name][i], ref, 2e-4, 2e-4 * float(ref.abs().max()), f"{name} vs ops")
    again = model.get_score_raw(rep, b, poses[0])
    assert torch.equal(again, outs["two_launches"][0]), "two runs differ"


def test_get_score_fused_falls_back_when_unsupported(dev):
    """Molecules above the kernels' size (32 atoms) take the operator path."""
    import numpy as np
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import escore, plan as P
    from moleculesde_amd.batch import Batch
    from moleculesde_amd.synthetic import make_molecule
    rng = np.random.default_rng(6)
    b = G.prepare_batch(Batch.from_data_list([make_molecule(rng, n) for n in (8, 40)]), dev)
    model = G.SDEModel2Dto3D_02(emb_dim=32, hidden_dim=32, beta_min=0.1, beta_max=1.0, num_diffusion_timesteps=1000,
                                beta_schedule=None, SDE_type="VE", use_extend_graph=True).to(dev).eval()
    assert not escore.score_supported(model, P.get_plan(b))
    gnn = G.GNN(3, 32, gnn_type="GIN").to(dev).eval()
    with torch.no_grad():
        rep = gnn(b.x, b.edge_index, b.edge_attr)
    out = model.get_score_raw(rep, b, b.positions)
    assert out.shape == (48, 3) and torch.isfinite(out).all()
