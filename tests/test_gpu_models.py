"""Model-level parity (GPU box): the HIP-backed classes of moleculesde_amd.geom3d against
(a) the committed golden vectors produced by the reference's own files and (b) the oracle
(oracle/restate.py) on seeded inputs at the BASELINE.json sizes, plus size-independent properties
(rotation equivariance, molecule-permutation invariance) at full size."""
import numpy as np
import pytest
import torch
from moleculesde_amd import slabs, wcache  # noqa: E402

pytestmark = pytest.mark.gpu

from oracle import restate as R  # noqa: E402
from helpers import assert_close, batch_from, disable_dropout, load_golden, record_parity, sub  # noqa: E402

TOY = dict(emb=16, filters=16, interactions=2, gaussians=51)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from moleculesde_amd import _lib
    _lib.load()
    return torch.device("cuda", 0)


def _grads_close(model, ggrads, rtol, atol_scale, what):
    scale = max(float(v.abs().max()) for v in ggrads.values())
    for n, p in model.named_parameters():
        if n in ggrads:
            assert p.grad is not None, f"{what}: no grad for {n}"
            assert_close(p.grad, ggrads[n], rtol, atol_scale * scale, f"{what}:{n}")


def _grads_close_l2(model, ggrads, rel_l2, max_abs_scale, what):
    """Deep-network gradients at full size: a handful of ReLU gates sit within rounding of zero and
    flip between the CPU and GPU forward, so individual entries can move by ~1e-2 of the scale while the
    tensor as a whole agrees.  Bar: relative L2 error of every parameter gradient <= rel_l2 (of the
    larger of its own norm and 1e-3 of the model's largest gradient norm) and no entry off by more
    than max_abs_scale * (largest gradient magnitude of the model)."""
    scale = max(float(v.abs().max()) for v in ggrads.values())
    nmax = max(float(v.double().norm()) for v in ggrads.values())
    for n, p in model.named_parameters():
        if n in ggrads:
            assert p.grad is not None, f"{what}: no grad for {n}"
            ref = ggrads[n].double().cpu()
            got = p.grad.double().cpu()
            err = float((got - ref).norm())
            den = max(float(ref.norm()), 1e-3 * nmax)
            assert err <= rel_l2 * den, f"{what}:{n}: rel L2 {err / den:.3e}"
            assert float((got - ref).abs().max()) <= max_abs_scale * scale, f"{what}:{n}: max abs"


def _s23(mod, E):
    return disable_dropout(mod.SDEModel2Dto3D_02(emb_dim=E, hidden_dim=32, beta_min=0.2, beta_max=1.0,
                                                 num_diffusion_timesteps=1000, beta_schedule=None, SDE_type="VE",
                                                 use_extend_graph=True))


# ---------------------------------------------------------------------------- golden fixtures ---
def test_golden_toy_gnn(dev):
    import moleculesde_amd.geom3d as G
    g = load_golden("toy_gnn.npz")
    b = G.prepare_batch(batch_from(g), dev)
    m = G.GNN(3, TOY["emb"], JK="last", drop_ratio=0, gnn_type="GIN")
    m.load_state_dict(sub(g, "sd."))
    m.to(dev).train()
    out = m(b.x, b.edge_index, b.edge_attr)
    assert_close(out, g["out"], 1e-4, 1e-5, "gnn out")
    out.pow(2).sum().backward()
    _grads_close(m, sub(g, "grad."), 1e-3, 1e-4, "gnn")
    for k, v in sub(g, "sd_after.").items():
        assert_close(m.state_dict()[k], v, 1e-4, 1e-5, "sd_after." + k)
    # the reference's other call form: forward(data)
    m.zero_grad()
    assert_close(m(b), out, 1e-6, 1e-6, "forward(data) form")


def test_golden_toy_schnet(dev):
    import moleculesde_amd.geom3d as G
    g = load_golden("toy_schnet.npz")
    b = G.prepare_batch(batch_from(g), dev)
    m = G.SchNet(hidden_channels=TOY["emb"], num_filters=TOY["filters"], num_interactions=TOY["interactions"],
                 num_gaussians=TOY["gaussians"], cutoff=10, readout="mean", node_class=119)
    m.load_state_dict(sub(g, "sd."))
    m.to(dev)
    out, h = m(b.x[:, 0], b.positions, b.batch, return_latent=True)
    assert_close(out, g["out"], 1e-4, 1e-5, "schnet out")
    assert_close(h, g["h"], 1e-4, 1e-5, "schnet h")
    (h.pow(2).sum() + out.sum()).backward()
    _grads_close(m, sub(g, "grad."), 1e-3, 1e-4, "schnet")


def test_golden_toy_sde2d3d(dev):
    import moleculesde_amd.geom3d as G
    g = load_golden("toy_sde2d3d.npz")
    b = G.prepare_batch(batch_from(g), dev)
    m = _s23(G, TOY["emb"])
    m.load_state_dict(sub(g, "sd."))
    m.to(dev).train()
    m.noise = G.CpuReplayNoise(int(g["seed"]))
    h2 = torch.from_numpy(g["h2"]).to(dev).requires_grad_(True)
    loss = m(h2, b, anneal_power=0)["position"]
    assert_close(loss, g["loss"], 1e-4, 1e-5, "loss 2d3d")
    loss.backward()
    assert_close(h2.grad, g["grad_h2"], 1e-3, 1e-4 * float(np.abs(g["grad_h2"]).max()), "grad h2")
    _grads_close(m, sub(g, "grad."), 1e-3, 2e-4, "sde2d3d")
    for k, v in sub(g, "sd_after.").items():
        assert_close(m.state_dict()[k], v, 1e-4, 1e-5, "sd_after." + k)
    m.eval()
    score = m.get_score(torch.from_numpy(g["h2"]).to(dev), b, torch.from_numpy(g["gs_pos"]).to(dev), None,
                        torch.from_numpy(g["gs_t_pos"]).to(dev))
    assert_close(score, g["gs_score"], 1e-3, 1e-4, "get_score")


def test_golden_qm9_schnet_config1(dev):
    """BASELINE.json configs[0] shape on the GPU: both the decomposed and (F=128 only) fused paths."""
    import moleculesde_amd.geom3d as G
    g = load_golden("qm9_schnet.npz")
    b = G.prepare_batch(batch_from(g), dev)
    m = G.SchNet(hidden_channels=32, num_filters=32, num_interactions=3, num_gaussians=51, cutoff=10,
                 readout="mean", node_class=119)
    m.load_state_dict(sub(g, "sd."))
    m.to(dev)
    out, h = m(b.x, b.positions, b.batch, return_latent=True)
    assert_close(out, g["out"], 1e-4, 1e-5, "qm9 out")
    assert_close(h, g["h"], 1e-4, 1e-5, "qm9 h")
    with torch.no_grad():
        out2, _ = m(b.x, b.positions, b.batch, return_latent=True)
    assert_close(out2, g["out"], 1e-4, 1e-5, "qm9 out no_grad")


def test_golden_sampler(dev):
    """5 predictor-corrector steps of the intended 2D->3D sampler with replayed noise."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import sampler
    g = load_golden("sampler.npz")
    b = G.prepare_batch(batch_from(g), dev)
    m = _s23(G, TOY["emb"])
    m.load_state_dict(sub(g, "sd."))
    m.to(dev).eval()
    rep = torch.from_numpy(g["rep"]).to(dev)
    pos = torch.from_numpy(g["pos0"]).to(dev)
    n = pos.size(0)
    for i, tval in enumerate(torch.from_numpy(g["ts"])):
        vec_t = torch.ones(n, device=dev) * float(tval)
        pos, _ = sampler.corrector_update(m.sde_pos, m, rep, b, pos, vec_t, float(g["snr"]), float(g["scale_eps"]), 1,
                                          noises=[torch.from_numpy(g["noise_corr"][i]).to(dev)])
        pos, _ = sampler.predictor_update(m.sde_pos, m, rep, b, pos, vec_t, noise=torch.from_numpy(g["noise_pred"][i]).to(dev))
        assert_close(pos, g["traj"][i], 1e-3, 1e-4, f"sampler step {i}")


# ------------------------------------------------------------- oracle at BASELINE sizes (bs 256) ---
def _pair_models(dev, seed=0):
    """Oracle (CPU) and product (GPU) models at the README configuration with identical weights."""
    import moleculesde_amd.geom3d as G
    torch.manual_seed(seed)
    om = R.build_models(use_3d2d=False)
    disable_dropout(om["SDE_2Dto3D_model"])
    pm = {
        "model_2D": G.GNN(5, 300, JK="last", drop_ratio=0, gnn_type="GIN"),
        "model_3D": G.SchNet(hidden_channels=300, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=10,
                             readout="mean", node_class=119),
        "SDE_2Dto3D_model": _s23(G, 300),
    }
    for k in pm:
        pm[k].load_state_dict(om[k].state_dict())
        pm[k].to(dev).train()
        om[k].train()
    return om, pm


def _oracle_step(om, cpu_b, seed, dtype=torch.float32):
    """configs[1] losses + backward on the oracle; noise is always drawn in fp32 (program order)."""
    orig = torch.randn_like
    torch.randn_like = lambda x, **k: orig(x.float()).to(x.dtype)
    try:
        b = cpu_b.clone()
        b.positions = b.positions.to(dtype)
        torch.manual_seed(seed)
        h2 = om["model_2D"](b.x, b.edge_index, b.edge_attr)
        _, h3 = om["model_3D"](b.x[:, 0], b.positions, b.batch, return_latent=True)
        cl, _ = R.dual_CL(h2, h3, 0.1)
        l23 = om["SDE_2Dto3D_model"](h2, b, anneal_power=0)["position"]
        (cl + l23).backward()
    finally:
        torch.randn_like = orig
    return h2.detach(), h3.detach(), cl.detach(), l23.detach()


def test_bs256_forward_backward_vs_oracle(dev):
    """configs[1]: GIN + SchNet + contrastive + SDE2Dto3D_02 VE, emb 300, bs 256.
    Forward (node representations, both losses): fp32 tolerance 2e-4 of the tensor scale / 1e-3
    relative on the losses, against the fp32 oracle.
    Gradients: GIN's 10 ReLU+BatchNorm stages make a few gates flip under ANY fp32 re-association --
    the fp32 oracle itself differs from an fp64 run of the same oracle by ~8e-3 relative L2 on GIN
    gradients (measured in this test).  So gradients are judged against the fp64 oracle, and the HIP
    path is held to FIXED bars per model (round 6; before: max(3 x err_cpu32, 2e-3), which floated with the oracle's own
    error): relative L2 against the fp64 oracle <= 6e-3 for GIN (measured 2.44e-3 on MI355X, the CPU fp32 oracle itself
    2.25e-3: profiles/r06_parity_numbers.txt), <= 1e-4 for SchNet (measured 1.7e-6), <= 2e-3 for the 2D->3D model (3.1e-4)."""
    import copy
    import moleculesde_amd.geom3d as G
    from moleculesde_amd.synthetic import make_batch
    from moleculesde_amd import pretrain
    om, pm = _pair_models(dev)
    om64 = {k: copy.deepcopy(m).double() for k, m in om.items()}
    cpu_b = make_batch(256, seed=0)
    dev_b = G.prepare_batch(cpu_b.clone(), dev)
    h2o, h3o, clo, l23o = _oracle_step(om, cpu_b, 123)
    _oracle_step(om64, cpu_b, 123, torch.float64)
    noise = G.CpuReplayNoise(123)
    pm["SDE_2Dto3D_model"].noise = noise
    args = pretrain.readme_args()
    h2 = pm["model_2D"](dev_b.x, dev_b.edge_index, dev_b.edge_attr)
    _, h3 = pm["model_3D"](dev_b.x[:, 0], dev_b.positions, dev_b.batch, return_latent=True)
    cl, _ = pretrain.dual_CL(h2, h3, args, noise)
    l23 = pm["SDE_2Dto3D_model"](h2, dev_b, anneal_power=0)["position"]
    (cl + l23).backward()
    assert_close(h2, h2o, 2e-4, 2e-4 * float(h2o.abs().max()), "GIN node repr")
    assert_close(h3, h3o, 2e-4, 2e-4 * float(h3o.abs().max()), "SchNet node repr")
    assert_close(cl, clo, 1e-3, 0, "contrastive loss")
    assert_close(l23, l23o, 1e-3, 0, "2D->3D loss")
    for k in pm:
        g64 = {n: p.grad for n, p in om64[k].named_parameters() if p.grad is not None}
        g32 = {n: p.grad for n, p in om[k].named_parameters() if p.grad is not None}
        nmax = max(float(v.norm()) for v in g64.values())
        e_hip, e_cpu = {}, {}
        for n, p in pm[k].named_parameters():
            if n not in g64:
                continue
            assert p.grad is not None, f"{k}:{n} has no gradient"
            den = max(float(g64[n].norm()), 1e-3 * nmax)
            if g64[n].numel() == 1:
                # GINConv.eps: d/d(eps) = sum_i g_i.x_i over ~1e6 terms that BatchNorm makes cancel almost
                # completely (the MLP output is nearly invariant to the scale of its input); the scalar is
                # ill-conditioned, so it is measured against the model's gradient scale instead of itself
                den = max(den, 5e-2 * nmax)
            e_hip[n] = float((p.grad.double().cpu() - g64[n]).norm()) / den
            e_cpu[n] = float((g32[n].double() - g64[n]).norm()) / den
        bar = {"model_2D": 6e-3, "model_3D": 1e-4, "SDE_2Dto3D_model": 2e-3}[k]
        worst = max(e_hip, key=e_hip.get)
        record_parity(f"bs256 gradients, {k}: worst rel-L2 vs fp64 oracle: hip {e_hip[worst]:.2e} ({worst}); "
                      f"cpu fp32 oracle worst {max(e_cpu.values()):.2e}; fixed bar {bar:.1e}")
        assert e_hip[worst] <= bar, f"{k}:{worst}: rel-L2 vs fp64 oracle {e_hip[worst]:.2e} > {bar:.2e}"


def test_loss_curve_vs_oracle(dev):
    """10 Adam steps (README learning rates) at bs 64, emb 300 with replayed noise, product Trainer (flat HIP Adam) vs
    oracle torch.optim.Adam.  FIXED bars (round 6; before: max(1e-3, 3 x the oracle's own fp32-vs-fp64 distance), which
    floated): the first three steps within 1e-3 relative (BASELINE.json's target; measured 3.5e-7 / 1.1e-4 / 1.5e-4), the
    first five within 2e-3 (8.1e-4), all ten within 2e-2 (8.8e-3).  A flat 1e-3 over ten steps does NOT hold at this size and
    is not a property of the kernels: the training dynamics amplify rounding, and the fp32 CPU oracle itself ends 5.3e-3 away
    from an fp64 run of the same oracle (profiles/r06_parity_numbers.txt).  The reference-generated 20-step curve (bs 8,
    tests/golden/losscurve.npz) IS held to a flat 1e-3: test_golden_losscurve_through_hip_trainer."""
    import copy
    import moleculesde_amd.geom3d as G
    from moleculesde_amd.synthetic import make_batch
    from moleculesde_amd import pretrain
    args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=0)
    torch.manual_seed(1)
    tr = pretrain.Trainer(args, dev)
    disable_dropout(tr.models["SDE_2Dto3D_model"])
    om = R.build_models(use_3d2d=False)
    disable_dropout(om["SDE_2Dto3D_model"])
    for k in om:
        om[k].load_state_dict(tr.models[k].state_dict())
        om[k].train()
    om64 = {k: copy.deepcopy(m).double() for k, m in om.items()}
    opt = R.make_optimizer(om, lr=1e-4, gnn_2d_lr_scale=1.0, gnn_3d_lr_scale=0.1)
    opt64 = R.make_optimizer(om64, lr=1e-4, gnn_2d_lr_scale=1.0, gnn_3d_lr_scale=0.1)
    cpu_b = make_batch(64, seed=2)
    b64 = cpu_b.clone()
    b64.positions = b64.positions.double()
    dev_b = G.prepare_batch(cpu_b.clone(), dev)
    orig = torch.randn_like
    ref, ref64, got = [], [], []
    for step in range(10):
        for models, o, bb, acc in ((om, opt, cpu_b, ref), (om64, opt64, b64, ref64)):
            torch.randn_like = lambda x, **k: orig(x.float()).to(x.dtype)
            try:
                torch.manual_seed(500 + step)
                loss, _ = R.pretrain_losses(models, bb, T=0.1, coeff_3d2d=0.0)
            finally:
                torch.randn_like = orig
            o.zero_grad()
            loss.backward()
            o.step()
            acc.append(loss.item())
        tr.noise = G.CpuReplayNoise(500 + step)
        tr.models["SDE_2Dto3D_model"].noise = tr.noise
        l, _ = tr.step(dev_b)
        got.append(float(l))
    got, ref, ref64 = np.array(got), np.array(ref), np.array(ref64)
    rel = np.abs(got - ref) / np.abs(ref)
    noise_floor = np.abs(ref - ref64) / np.abs(ref64)
    record_parity("loss curve (10 Adam steps, bs 64, emb 300): rel err per step vs fp32 oracle " +
                  " ".join("%.1e" % v for v in rel) + "; max %.2e; fp32-vs-fp64 oracle max %.2e" % (rel.max(), noise_floor.max()))
    assert rel[:3].max() <= 1e-3, rel                       # before the dynamics amplify rounding
    assert rel[:5].max() <= 2e-3, rel
    assert rel.max() <= 2e-2, (rel, noise_floor, got, ref)


# ------------------------------------------------------------------ properties at full size ------
def test_fullsize_properties(dev):
    """bs 256, emb 300: (1) the 2D->3D score rotates with the input (SE(3) frame), translation of a
    whole molecule is NOT an invariance of this model (frames use absolute positions) so only rotation
    about the origin is checked; (2) SchNet is invariant to rotation+translation; (3) permuting the
    molecules of the batch permutes the outputs; (4) two launches are bitwise identical."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd.batch import Batch
    from moleculesde_amd.synthetic import make_batch, make_molecule
    torch.manual_seed(3)
    sch = G.SchNet(hidden_channels=300, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=10,
                   readout="mean", node_class=119).to(dev)
    gnn = G.GNN(5, 300, JK="last", drop_ratio=0, gnn_type="GIN").to(dev).eval()
    s23 = _s23(G, 300).to(dev).eval()
    b = G.prepare_batch(make_batch(256, seed=4), dev)
    with torch.no_grad():
        h2 = gnn(b.x, b.edge_index, b.edge_attr)
        out, h = sch(b.x[:, 0], b.positions, b.batch, return_latent=True)
        out_b, h_b = sch(b.x[:, 0], b.positions, b.batch, return_latent=True)
        assert torch.equal(h, h_b) and torch.equal(out, out_b)               # (4) deterministic
        Q, _ = torch.linalg.qr(torch.randn(3, 3, device=dev))
        if torch.det(Q) < 0:
            Q[:, 0] = -Q[:, 0]
        shift = torch.randn(256, 3, device=dev)[b.batch]
        _, h_rt = sch(b.x[:, 0], b.positions @ Q.T + shift, b.batch, return_latent=True)
        assert_close(h_rt, h, 1e-3, 1e-3 * float(h.abs().max()), "SchNet SE(3) invariance")       # (2)
        t_pos = torch.full((b.x.size(0),), 0.5, device=dev)
        pos_p = b.positions + 0.2 * torch.randn_like(b.positions)
        sc = s23.get_score(h2, b, pos_p, None, t_pos)
        sc_r = s23.get_score(h2, b, pos_p @ Q.T, None, t_pos)
        assert_close(sc_r, sc @ Q.T, 5e-3, 5e-3 * float(sc.abs().max()), "score rotation equivariance")  # (1)
    # (3) molecule permutation
    rng = np.random.default_rng(7)
    mols = [make_molecule(rng) for _ in range(64)]
    perm = rng.permutation(64)
    b1 = G.prepare_batch(Batch.from_data_list(mols), dev)
    b2 = G.prepare_batch(Batch.from_data_list([mols[i] for i in perm]), dev)
    with torch.no_grad():
        o1 = sch(b1.x[:, 0], b1.positions, b1.batch)
        o2 = sch(b2.x[:, 0], b2.positions, b2.batch)
    assert_close(o2, o1[torch.from_numpy(perm).to(dev)], 1e-5, 1e-5, "molecule permutation")


def test_fused_vs_decomposed_fullsize(dev):
    """SchNet forward at bs 256 / emb 300: fused MFMA CFConv path (no_grad) == decomposed path."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd.synthetic import make_batch
    torch.manual_seed(5)
    sch = G.SchNet(hidden_channels=300, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=10,
                   readout="mean", node_class=119).to(dev)
    b = G.prepare_batch(make_batch(256, seed=6), dev)
    _, h_dec = sch(b.x[:, 0], b.positions, b.batch, return_latent=True)
    with torch.no_grad():
        _, h_fused = sch(b.x[:, 0], b.positions, b.batch, return_latent=True)
    assert_close(h_fused, h_dec.detach(), 1e-4, 1e-4 * float(h_dec.abs().max()), "fused vs decomposed SchNet")


def test_empty_and_ragged_inputs(dev):
    """Edge cases: single-atom molecules (no radius/bond edges), 2-atom molecules, one big molecule."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd.batch import Batch, MolData, extend_graph_index
    from moleculesde_amd.synthetic import make_molecule
    rng = np.random.default_rng(9)
    lone = MolData(x=torch.tensor([[5, 0, 0, 0, 0, 0, 0, 0, 0]]), edge_index=torch.zeros(2, 0, dtype=torch.long),
                   edge_attr=torch.zeros(0, 3, dtype=torch.long), positions=torch.zeros(1, 3))
    lone.extended_edge_index = extend_graph_index(lone.edge_index, 1)
    mols = [lone, make_molecule(rng, 2), make_molecule(rng, 20), lone, make_molecule(rng, 7)]
    cpu_b = Batch.from_data_list(mols)
    b = G.prepare_batch(cpu_b.clone(), dev)
    torch.manual_seed(0)
    osch = R.SchNet(hidden_channels=32, num_filters=128, num_interactions=2, num_gaussians=51, cutoff=10, readout="mean", node_class=119)
    ognn = R.GNN(2, 32)
    sch = G.SchNet(hidden_channels=32, num_filters=128, num_interactions=2, num_gaussians=51, cutoff=10, readout="mean", node_class=119)
    gnn = G.GNN(2, 32, gnn_type="GIN")
    sch.load_state_dict(osch.state_dict()); gnn.load_state_dict(ognn.state_dict())
    sch.to(dev); gnn.to(dev)
    o_ref, h_ref = osch(cpu_b.x[:, 0], cpu_b.positions, cpu_b.batch, return_latent=True)
    o, h = sch(b.x[:, 0], b.positions, b.batch, return_latent=True)
    assert_close(h, h_ref.detach(), 1e-4, 1e-5, "ragged schnet h")
    assert_close(o, o_ref.detach(), 1e-4, 1e-5, "ragged schnet out")
    with torch.no_grad():
        _, hf = sch(b.x[:, 0], b.positions, b.batch, return_latent=True)
    assert_close(hf, h_ref.detach(), 1e-4, 1e-5, "ragged schnet fused")
    assert_close(gnn(b.x, b.edge_index, b.edge_attr), ognn(cpu_b.x, cpu_b.edge_index, cpu_b.edge_attr).detach(), 1e-4, 1e-5, "ragged gnn")


@pytest.mark.parametrize("adam_outside", [False, True])
def test_hipgraph_step_matches_eager(dev, adam_outside):
    """The captured hipGraph step (fwd + bwd + grad flattening + flat Adam) replays the same arithmetic
    as the eager step: same loss on the same weights with dropout off, and parameters move identically."""
    import copy
    import moleculesde_amd.geom3d as G
    from moleculesde_amd.synthetic import make_batch
    from moleculesde_amd import pretrain
    args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=0, emb_dim=64)
    torch.manual_seed(3)
    tr_e = pretrain.Trainer(args, dev)
    tr_g = pretrain.Trainer(args, dev)
    tr_g.adam_outside_graph = adam_outside      # True = the multi-GPU structure: graph, all-reduce, Adam
    for k in tr_e.models:
        tr_g.models[k].load_state_dict(tr_e.models[k].state_dict())
        disable_dropout(tr_e.models[k]); disable_dropout(tr_g.models[k])
    b = G.prepare_batch(make_batch(32, seed=8), dev)

    class FixedNoise(G.DeviceNoise):
        def __init__(self, n, B):
            g = torch.Generator().manual_seed(5)
            self.z = torch.randn(n, 3, generator=g).to(dev)
            self.t = torch.randint(0, 1000, (B // 2 + 1,), generator=g).to(dev)
            self.p = torch.randperm(n, generator=g).to(dev)
        def randn_like(self, x): return self.z.clone()
        def randint(self, high, size, device): return self.t.clone()
        def randperm(self, n, device): return self.p.clone()
    for tr in (tr_e, tr_g):
        tr.noise = FixedNoise(b.x.size(0), 32)
        tr.models["SDE_2Dto3D_model"].noise = tr.noise
    tr_g.step(b); tr_e.step(b)                       # eager warm-up on both (identical)
    tr_g.capture(b)
    for _ in range(3):
        le, _ = tr_e.step(b)
        lg = tr_g.step_graph(b)
        assert_close(lg, le, 1e-5, 1e-6, "graph vs eager loss")
    # parameters whose gradient is analytically zero (bias before BatchNorm, key bias under softmax) carry
    # pure rounding noise that Adam normalises to +-lr, so agreement is judged on the whole vector
    d = float((tr_g.opt.flat_p - tr_e.opt.flat_p).norm() / tr_e.opt.flat_p.norm())
    assert d < 1e-4, d


# ------------------------------------------------------------------ 3D -> 2D dense head (a12-a14) ---
def _s32(mod, E):
    return mod.SDEModel3Dto2D_node_adj_dense(dim3D=E, c_init=2, c_hid=8, c_final=4, num_heads=4, adim=16, nhid=16,
                                             num_layers=4, emb_dim=E, num_linears=3, beta_min=0.1, beta_max=1.0,
                                             num_diffusion_timesteps=1000, SDE_type="VE", num_class_X=119,
                                             noise_on_one_hot=True)


@pytest.mark.parametrize("mode", ["hip", "auto"])
def test_golden_dense_head_genuine(dev, mode):
    """Edge / node score networks vs the golden produced by the GENUINE reference import (no stand-ins);
    mode 'hip' runs every Linear (incl. the node MLP chain) on the hand-written MFMA kernels."""
    from moleculesde_amd import hip
    from moleculesde_amd.geom3d import sde_3d_to_2d as S
    hip.set_linear_mode(mode)
    try:
        g = load_golden("dense_head.npz")
        edge = S.EdgeScoreNetwork_dense(dim3D=12, nhid=8, num_layers=3, num_linears=3, c_init=2, c_hid=4, c_final=2,
                                        adim=8, num_heads=4, conv="MLP")
        node = S.NodeScoreNetwork_dense(nfeat=12, depth=3, nhid=8, nout=7)
        edge.load_state_dict(sub(g, "edge.sd."))
        node.load_state_dict(sub(g, "node.sd."))
        edge.to(dev); node.to(dev)
        x = torch.from_numpy(g["x"]).to(dev).requires_grad_(True)
        a = torch.from_numpy(g["adj"]).to(dev).requires_grad_(True)
        flags = torch.from_numpy(g["flags"]).to(dev)
        se, sn = edge(x, a, flags), node(x, a, flags)
        assert_close(se, g["score_edge"], 1e-4, 1e-5, "edge score")
        assert_close(sn, g["score_node"], 1e-4, 1e-5, "node score")
        (se.pow(2).sum() + sn.pow(2).sum()).backward()
        assert_close(x.grad, g["grad_x"], 1e-3, 1e-4, "grad x")
        assert_close(a.grad, g["grad_adj"], 1e-3, 1e-4, "grad adj")
        _grads_close(edge, sub(g, "edge.grad."), 1e-3, 1e-4, "edge net")
        _grads_close(node, sub(g, "node.grad."), 1e-3, 1e-4, "node net")
    finally:
        hip.set_linear_mode("auto")


def test_golden_toy_sde3d2d(dev):
    import moleculesde_amd.geom3d as G
    g = load_golden("toy_sde3d2d.npz")
    b = G.prepare_batch(batch_from(g), dev)
    m = _s32(G, TOY["emb"])
    m.load_state_dict(sub(g, "sd."))
    m.to(dev).train()
    m.noise = G.CpuReplayNoise(int(g["seed"]))
    h3 = torch.from_numpy(g["h3"]).to(dev).requires_grad_(True)
    from moleculesde_amd.geom3d import dense_head
    calls = dense_head.FUSED_CALLS
    lx, la = m(h3, b, reduce_mean=True, continuous=True, train=True, anneal_power=0)
    assert dense_head.FUSED_CALLS == calls + 1, "the fused head kernels (not the operator path on vendor GEMMs) must have run"
    assert_close(lx, g["loss_x"], 1e-4, 1e-6, "loss_x")
    assert_close(la, g["loss_adj"], 1e-4, 1e-6, "loss_adj")
    (lx + la).backward()
    assert_close(h3.grad, g["grad_h3"], 1e-3, 1e-4 * float(np.abs(g["grad_h3"]).max()), "grad h3")
    _grads_close(m, sub(g, "grad."), 1e-3, 2e-4, "sde3d2d")


def test_sde3d2d_outside_the_fused_head_raises_unless_allowed(dev):
    """A head configuration the kernels do not cover (nhid = 8) must not drop to the operator path (vendor batched GEMMs)
    silently: it raises, and runs operator by operator only with `allow_operator_path = True`."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd._lib import MsdeHipError
    from moleculesde_amd.synthetic import make_batch
    b = G.prepare_batch(make_batch(4, seed=2), dev)
    m = G.SDEModel3Dto2D_node_adj_dense(dim3D=16, c_init=2, c_hid=8, c_final=4, num_heads=4, adim=16, nhid=8, num_layers=4,
                                        emb_dim=16, num_linears=3, beta_min=0.1, beta_max=1.0, num_diffusion_timesteps=1000,
                                        SDE_type="VE", num_class_X=119, noise_on_one_hot=True).to(dev).train()
    h3 = torch.randn(b.x.size(0), 16, device=dev)
    with pytest.raises(MsdeHipError, match="allow_operator_path"):
        m(h3, b, reduce_mean=True, continuous=True, train=True, anneal_power=0)
    m.allow_operator_path = True
    lx, la = m(h3, b, reduce_mean=True, continuous=True, train=True, anneal_power=0)
    assert torch.isfinite(lx) and torch.isfinite(la)


def test_golden_f3_sde3d2d_02(dev):
    """§8 f3: SDEModel3Dto2D_node_adj_dense_02 (concatenated embeddings, 2 * dim3D score networks) against the fixture
    produced by the reference's own class (oracle/make_golden_f3_02.py): state-dict keys, both losses under replayed
    noise, the gradient of the 3D representation and every parameter gradient."""
    import moleculesde_amd.geom3d as G
    g = load_golden("f3_sde3d2d_02.npz")
    b = G.prepare_batch(batch_from(g), dev)
    E = g["h3"].shape[1]
    m = G.SDEModel3Dto2D_node_adj_dense_02(dim3D=E, c_init=2, c_hid=8, c_final=4, num_heads=4, adim=16, nhid=16,
                                           num_layers=4, emb_dim=E, num_linears=3, beta_min=0.1, beta_max=1.0,
                                           num_diffusion_timesteps=1000, SDE_type="VE", num_class_X=119,
                                           noise_on_one_hot=True)
    assert [k for k, _ in m.named_parameters()] == list(g["param_names"])
    _set_parameters_like_generator(m, 4200)
    m.to(dev).train()
    m.noise = G.CpuReplayNoise(int(g["seed"]))
    h3 = torch.from_numpy(g["h3"]).to(dev).requires_grad_(True)
    from moleculesde_amd.geom3d import dense_head
    calls = dense_head.FUSED_CALLS
    lx, la = m(h3, b, reduce_mean=True, continuous=True, train=True, anneal_power=0)
    assert dense_head.FUSED_CALLS == calls + 1, "the fused head kernels (not the operator path) must have run"
    assert_close(lx, g["loss_x"], 1e-4, 1e-6, "loss_x")
    assert_close(la, g["loss_adj"], 1e-4, 1e-6, "loss_adj")
    (lx + la).backward()
    assert_close(h3.grad, g["grad_h3"], 1e-3, 1e-4 * float(np.abs(g["grad_h3"]).max()), "grad h3")
    _grads_close(m, sub(g, "grad."), 1e-3, 2e-4, "sde3d2d_02")


def test_golden_f4_painn(dev):
    """§8 f4: the PaiNN encoder (painn.py:118-269) on the kernel operator set against the fixture produced by the
    reference's own class (oracle/make_golden_painn.py): readout, latent atom features, the position gradient (force path)
    and every parameter gradient; the padding row of the embedding stays without gradient."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import dd
    g = load_golden("f4_painn.npz")
    b = G.prepare_batch(batch_from(g), dev)
    m = G.PaiNN(n_atom_basis=32, n_interactions=3, n_rbf=20, cutoff=float(g["cutoff"]), max_z=119, n_out=1, readout="mean")
    assert [k for k, _ in m.named_parameters()] == list(g["param_names"])
    _set_parameters_like_generator(m, 6100)
    with torch.no_grad():
        m.embedding.weight[0].zero_()
    m.to(dev)
    ei = torch.from_numpy(g["radius_edge_index"]).to(dev)
    pos = b.positions.clone().requires_grad_(True)
    dd.CALLS.clear()
    h, q = m(b.x[:, 0], pos, ei, b.batch, return_latent=True)
    assert_close(h, g["h"], 1e-4, 1e-5, "readout")
    assert_close(q, g["q"], 1e-4, 1e-5, "latent")
    (h.pow(2).sum() + q.sum()).backward()
    assert_close(pos.grad, g["grad_pos"], 1e-3, 1e-4 * float(np.abs(g["grad_pos"]).max()), "grad positions")
    _grads_close(m, sub(g, "grad."), 1e-3, 2e-4, "painn")
    assert float(m.embedding.weight.grad[0].abs().max()) == 0.0
    assert dd.CALLS.get("msde_cfconv_aggregate", 0) >= 3 * 7 - 3 and dd.CALLS.get("msde_gemm_ex", 0) > 30, dd.CALLS


def test_painn_force_path_double_backward(dev):
    """PaiNN under the MD17 objective (finetune_MD17.py:47-78): forces by autograd.grad(create_graph=True), then a
    backward pass through them.  The operator set is closed under differentiation, so this runs on the same kernels;
    checked against central finite differences of the force-matching loss w.r.t. two parameters."""
    import moleculesde_amd.geom3d as G
    g = load_golden("f4_painn.npz")
    b = G.prepare_batch(batch_from(g), dev)
    m = G.PaiNN(n_atom_basis=32, n_interactions=2, n_rbf=20, cutoff=float(g["cutoff"]), max_z=119, n_out=1, readout="mean")
    _set_parameters_like_generator(m, 6100)
    m.to(dev)
    ei = torch.from_numpy(g["radius_edge_index"]).to(dev)
    f_t = torch.randn(b.x.size(0), 3, device=dev, generator=torch.Generator(device=dev).manual_seed(1))

    def loss_of():
        pos = b.positions.clone().requires_grad_(True)
        e = m(b.x[:, 0], pos, ei, b.batch).sum(dim=1, keepdim=True)
        force = -torch.autograd.grad(e, pos, grad_outputs=torch.ones_like(e), create_graph=True)[0]
        return (force - f_t).pow(2).mean()
    loss = loss_of()
    loss.backward()
    for name, idx in (("interactions.0.interatomic_context_net.1.weight", (5, 7)), ("filter_net.weight", (40, 3))):
        p = dict(m.named_parameters())[name]
        an = float(p.grad[idx])
        eps = 2e-3
        def bump(v):
            with torch.no_grad():
                p[idx] += v
        bump(eps)
        lp = float(loss_of())
        bump(-2 * eps)
        lm = float(loss_of())
        bump(eps)
        fd = (lp - lm) / (2 * eps)
        assert abs(an - fd) <= 5e-2 * max(abs(fd), abs(an)) + 1e-5, (name, an, fd)


def test_pretrain_step_with_painn_encoder(dev):
    """--model_3d PaiNN (pretrain_MoleculeSDE.py:212-221): the trainer builds the radius graph with the radius kernels,
    runs contrastive + 2D->3D losses over the PaiNN latent and updates every PaiNN parameter."""
    from moleculesde_amd import pretrain
    import moleculesde_amd.geom3d as G
    from moleculesde_amd.synthetic import make_batch
    args = pretrain.readme_args(model_3d="PaiNN", emb_dim=64, SDE_coeff_generative_3Dto2D=0)
    tr = pretrain.Trainer(args, dev)
    assert isinstance(tr.models["model_3D"], G.PaiNN)
    b = G.prepare_batch(make_batch(16, seed=5), dev)
    before = {k: v.detach().clone() for k, v in tr.models["model_3D"].named_parameters()}
    l0, parts = tr.step(b)
    for _ in range(3):
        l1, parts = tr.step(b)
    assert torch.isfinite(l0) and torch.isfinite(l1)
    assert b.radius_edge_index.size(0) == 2 and b.radius_edge_index.size(1) > 0
    moved = [k for k, v in tr.models["model_3D"].named_parameters() if not torch.equal(v, before[k])]
    assert len(moved) == len(before), set(before) - set(moved)


def test_bs256_full_pretrain_losses_vs_oracle(dev):
    """BASELINE.json configs[2] per-GPU work: all three losses (contrastive + 2D->3D + 3D->2D VE) at bs 256,
    emb 300, against the oracle with replayed noise: each loss term within 1e-3 relative."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd.synthetic import make_batch
    from moleculesde_amd import pretrain
    args = pretrain.readme_args()
    torch.manual_seed(4)
    tr = pretrain.Trainer(args, dev)
    disable_dropout(tr.models["SDE_2Dto3D_model"])
    om = R.build_models(use_3d2d=True)
    disable_dropout(om["SDE_2Dto3D_model"])
    for k in om:
        om[k].load_state_dict(tr.models[k].state_dict())
        om[k].train()
    cpu_b = make_batch(256, seed=3)
    dev_b = G.prepare_batch(cpu_b.clone(), dev)
    torch.manual_seed(77)
    loss_o, parts_o = R.pretrain_losses(om, cpu_b, T=0.1)
    tr.noise = G.CpuReplayNoise(77)
    tr.models["SDE_2Dto3D_model"].noise = tr.noise
    tr.models["SDE_3Dto2D_model"].noise = tr.noise
    loss, parts = tr.losses(dev_b)
    for k in ("CL", "2Dto3D", "3Dto2D"):
        assert_close(parts[k], parts_o[k].detach(), 1e-3, 0, f"loss term {k}")
    assert_close(loss, loss_o.detach(), 1e-3, 0, "total loss")
    loss.backward()
    loss_o.backward()
    g64n = {n: p.grad for n, p in om["SDE_3Dto2D_model"].named_parameters() if p.grad is not None}
    _grads_close_l2(tr.models["SDE_3Dto2D_model"], g64n, 5e-3, 2e-2, "3D->2D grads")


def test_sampler_graph_matches_eager(dev):
    """Config 4 loop: hipGraph replay of one predictor-corrector iteration == the eager loop, given the
    same noise (the device generator is re-seeded before each run)."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import sampler
    from moleculesde_amd.batch import Batch
    from moleculesde_amd.synthetic import make_molecule
    torch.manual_seed(0)
    rng = np.random.default_rng(1)
    mol = make_molecule(rng, 12)
    b = G.prepare_batch(Batch.from_data_list([mol] * 4), dev)
    gnn = G.GNN(3, 32, gnn_type="GIN").to(dev).eval()
    s23 = _s23(G, 32).to(dev).eval()
    with torch.no_grad():
        rep = gnn(b.x, b.edge_index, b.edge_attr)
    pos0 = torch.randn(b.x.size(0), 3, device=dev)
    outs = []
    for use_graph in (False, True):
        torch.manual_seed(123)
        torch.cuda.manual_seed(123)
        outs.append(sampler.position_PC_generation(s23, rep, b, num_steps=12, pos_init=pos0, use_graph=use_graph,
                                                   denoise=True))
    assert torch.isfinite(outs[0]).all()
    # identical kernels and (Philox) noise sequence; graph-mode RNG offsets advance identically
    assert_close(outs[1], outs[0], 1e-4, 1e-4, "graph vs eager sampler")


def test_inference_caches_follow_parameter_updates(dev):
    """The inference-time caches (concatenated parameters and their transposed copies, coordinate-independent score-network
    inputs) must not survive a parameter update: get_score after training steps == get_score of a FRESH model loaded with
    the same parameters (a training step's batched weight-copy refresh runs between the two evaluations)."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import pretrain
    from moleculesde_amd.synthetic import make_batch
    torch.manual_seed(0)
    args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=0, emb_dim=64)
    tr = pretrain.Trainer(args, dev)
    b = G.prepare_batch(make_batch(8, seed=3), dev)
    m = tr.models["SDE_2Dto3D_model"]
    gnn = tr.models["model_2D"]

    def score(model, enc):
        model.eval(); enc.eval()
        with torch.no_grad():
            rep = enc(b.x, b.edge_index, b.edge_attr)
            out = model.get_score(rep, b, b.positions, None, torch.full((b.x.size(0),), 0.5, device=dev))
        model.train(); enc.train()
        return out.clone()

    s0 = score(m, gnn)
    for _ in range(3):
        tr.step(b)
    s1 = score(m, gnn)
    assert (s1 - s0).abs().max() > 0, "the parameters did not move"
    args2 = pretrain.readme_args(SDE_coeff_generative_3Dto2D=0, emb_dim=64)
    fresh = pretrain.build_models(args2, dev)
    fresh["SDE_2Dto3D_model"].load_state_dict(m.state_dict())
    fresh["model_2D"].load_state_dict(gnn.state_dict())
    s2 = score(fresh["SDE_2Dto3D_model"], fresh["model_2D"])
    assert_close(s1, s2, 1e-5, 1e-6, "get_score after training steps vs a fresh model with the same parameters")


@pytest.mark.parametrize("sde_type", ["VE", "VP"])
def test_sampler_fused_arithmetic_matches_operator_path(dev, sde_type):
    """msde_pc_corrector / msde_pc_predictor (one kernel per half iteration, diffusion-time scalars tabulated once) against
    the operator-by-operator corrector_update / predictor_update on the same noise sequence; the inference-time caches
    (coordinate-independent features, parameter concatenations) are exercised by both runs."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import sampler
    from moleculesde_amd.batch import Batch
    from moleculesde_amd.synthetic import make_molecule
    torch.manual_seed(0)
    rng = np.random.default_rng(2)
    mol = make_molecule(rng, 13)
    b = G.prepare_batch(Batch.from_data_list([mol] * 3), dev)
    gnn = G.GNN(3, 32, gnn_type="GIN").to(dev).eval()
    bmin, bmax = (0.2, 1.0) if sde_type == "VE" else (0.2, 30.0)
    s23 = G.SDEModel2Dto3D_02(emb_dim=32, hidden_dim=32, beta_min=bmin, beta_max=bmax, num_diffusion_timesteps=1000,
                              beta_schedule=None, SDE_type=sde_type, use_extend_graph=True).to(dev).eval()
    with torch.no_grad():
        rep = gnn(b.x, b.edge_index, b.edge_attr)
    pos0 = torch.randn(b.x.size(0), 3, device=dev)
    outs = []
    keep = sampler.FUSED_PC
    try:
        for fused in (False, True):
            sampler.FUSED_PC = fused
            torch.manual_seed(7)
            torch.cuda.manual_seed(7)
            outs.append(sampler.position_PC_generation(s23, rep, b, num_steps=10, pos_init=pos0, use_graph=False, denoise=False,
                                                       torch_noise=True))
    finally:
        sampler.FUSED_PC = keep
    assert torch.isfinite(outs[0]).all()
    assert_close(outs[1], outs[0], 1e-4, 1e-5, "fused vs operator sampler arithmetic (%s)" % sde_type)
    # the production mode: iteration counter and noise inside the update kernels -- a trajectory is a function of its seed, and the
    # replayed hipGraph (no host work between iterations) walks the same trajectory as the host-launched loop
    kw = dict(num_steps=12, pos_init=pos0, denoise=False, noise_seed=1234)
    e1 = sampler.position_PC_generation(s23, rep, b, use_graph=False, **kw)
    e2 = sampler.position_PC_generation(s23, rep, b, use_graph=False, **kw)
    g1 = sampler.position_PC_generation(s23, rep, b, use_graph=True, **kw)              # 10 iterations in one graph launch
    g4 = sampler.position_PC_generation(s23, rep, b, use_graph=True, iters_per_graph=4, **kw)   # 2 x 4 + 2 single replays
    assert torch.equal(g4, g1)
    o1 = sampler.position_PC_generation(s23, rep, b, use_graph=False, **dict(kw, noise_seed=99))
    assert torch.isfinite(e1).all() and torch.equal(e1, e2)
    assert_close(g1, e1, 1e-5, 1e-6, "replayed vs host-launched trajectory")
    assert not torch.allclose(o1, e1)
    # the in-kernel draws are N(0,1): moments of what one predictor step adds (x - x_mean) / G over many atoms
    from moleculesde_amd import _lib, hip
    n = 20000
    z = torch.zeros(n, 3, device=dev)
    par = torch.tensor([[1.0, 1.0, 1.0, 1.0]], device=dev)
    x, xm = torch.empty_like(z), torch.empty_like(z)
    _lib.call("msde_pc_predictor", hip._p(z), hip._p(z), hip._p(None), hip._p(par), hip._p(None), 77, n, hip._p(x), hip._p(xm), hip._stream())
    d = (x - xm).flatten()
    assert abs(float(d.mean())) < 0.02 and abs(float(d.std()) - 1.0) < 0.02 and abs(float((d ** 4).mean()) - 3.0) < 0.15


@pytest.mark.parametrize("bs", [1, 4])
def test_md17_force_path_double_backward(dev, bs):
    """BASELINE.json config 5 (finetune_MD17.py:47-78): energy = Linear(SchNet(z, pos)), force =
    -d(energy)/d(pos) with create_graph=True, loss = L1(energy) + L1(force), backward through the forces.
    Energy, forces and every parameter gradient against the oracle's autograd on the CPU."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd.synthetic import make_md17_batch
    torch.manual_seed(7)
    kw = dict(hidden_channels=64, num_filters=32, num_interactions=3, num_gaussians=51, cutoff=10, readout="mean", node_class=119)
    osch = R.SchNet(**kw)
    ohead = torch.nn.Linear(64, 1)
    sch = G.SchNet(**kw)
    sch.load_state_dict(osch.state_dict())
    head = torch.nn.Linear(64, 1)
    head.load_state_dict(ohead.state_dict())
    sch.to(dev); head.to(dev)
    cpu_b = make_md17_batch(bs, seed=3, n_atoms=21)
    e_t = torch.randn(bs, 1)
    f_t = torch.randn(cpu_b.x.size(0), 3)

    def run(model, hd, b, et, ft):
        pos = b.positions.clone().requires_grad_(True)
        energy = hd(model(b.x, pos, b.batch))
        force = -torch.autograd.grad(energy, pos, grad_outputs=torch.ones_like(energy), create_graph=True,
                                     retain_graph=True)[0]
        loss = (energy - et).abs().mean() + (force - ft).abs().mean()
        loss.backward()
        return energy.detach(), force.detach(), loss.detach()

    eo, fo, lo = run(osch, ohead, cpu_b, e_t, f_t)
    dev_b = G.prepare_batch(cpu_b.clone(), dev)
    e, f, l = run(sch, head, dev_b, e_t.to(dev), f_t.to(dev))
    assert_close(e, eo, 1e-4, 1e-5, "energy")
    assert_close(f, fo, 1e-3, 1e-4 * float(fo.abs().max()), "force")
    assert_close(l, lo, 1e-4, 1e-6, "loss")
    gs = {n: p.grad for n, p in osch.named_parameters() if p.grad is not None}
    _grads_close(sch, gs, 2e-3, 2e-4, "SchNet double-backward grads")
    assert_close(head.weight.grad, ohead.weight.grad, 1e-3, 1e-5, "head grad")


@pytest.mark.timeout(600)
def test_md17_force_trainer_graph_replay_matches_eager_and_oracle(dev):
    """moleculesde_amd.finetune_md17.ForceTrainer (finetune_MD17.py:34-88): the whole force fine-tuning step -- energy,
    forces with create_graph, L1 losses with the 0.05 / 0.95 coefficients, backward through the forces, Adam -- captured
    as ONE hipGraph and replayed on new conformations gives the losses of the eager steps (same kernels: 1e-6) and of the
    reference loop on the oracle with torch.optim.Adam (1e-3 relative, BASELINE north star)."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import hip
    from moleculesde_amd.finetune_md17 import ForceTrainer
    from moleculesde_amd.synthetic import make_md17_batch
    kw = dict(hidden_channels=64, num_filters=32, num_interactions=3, num_gaussians=51, cutoff=10, readout="mean", node_class=119)
    bs, steps = 2, 5
    cpu_b = make_md17_batch(bs, seed=3, n_atoms=21)
    g = torch.Generator().manual_seed(12)
    confs = [cpu_b.positions + 0.05 * torch.randn(cpu_b.positions.shape, generator=g) for _ in range(steps)]
    ys = [torch.randn(bs, generator=g) for _ in range(steps)]
    fs = [torch.randn(cpu_b.x.size(0), 3, generator=g) for _ in range(steps)]

    torch.manual_seed(7)
    osch, ohead = R.SchNet(**kw), torch.nn.Linear(64, 1)
    state, hstate = {k: v.clone() for k, v in osch.state_dict().items()}, {k: v.clone() for k, v in ohead.state_dict().items()}
    oopt = torch.optim.Adam(list(osch.parameters()) + list(ohead.parameters()), lr=5e-4)
    ref = []
    for t in range(steps):          # the reference's loop, finetune_MD17.py:46-78
        pos = confs[t].clone().requires_grad_(True)
        e = ohead(osch(cpu_b.x, pos, cpu_b.batch)).squeeze(1)
        f = -torch.autograd.grad(e, pos, grad_outputs=torch.ones_like(e), create_graph=True, retain_graph=True)[0]
        loss = 0.05 * torch.nn.functional.l1_loss(e, ys[t]) + 0.95 * torch.nn.functional.l1_loss(f, fs[t])
        oopt.zero_grad(); loss.backward(); oopt.step()
        ref.append(float(loss))

    def make():
        sch, head = G.SchNet(**kw), torch.nn.Linear(64, 1)
        sch.load_state_dict(state); head.load_state_dict(hstate)
        return ForceTrainer(sch.to(dev), head.to(dev), lr=5e-4)
    b = G.prepare_batch(cpu_b.clone(), dev)
    ft = make()
    eager = []
    for t in range(steps):
        b.positions = confs[t].to(dev)
        eager.append(float(ft.step(b, ys[t].to(dev), fs[t].to(dev))))
    ft = make()
    p0 = ft.opt.flat_p.clone()
    b.positions = confs[0].to(dev)
    ft.capture(b, ys[0].to(dev), fs[0].to(dev))
    ft.opt.flat_p.copy_(p0); ft.opt.m.zero_(); ft.opt.v.zero_(); ft.opt.step_dev.zero_()
    wcache.bump_weight_epoch()
    replay = [float(ft.step_graph(confs[t].to(dev), ys[t].to(dev), fs[t].to(dev))) for t in range(steps)]
    print("MD17 losses: oracle", ref, "eager", eager, "graph", replay)
    for t in range(steps):
        assert abs(replay[t] - eager[t]) <= 1e-6 * abs(eager[t]) + 1e-7, (t, replay[t], eager[t])
        assert abs(eager[t] - ref[t]) <= 1e-3 * abs(ref[t]), (t, eager[t], ref[t])


def test_md17_force_trainer_frozen_biases(dev):
    """ADVICE r5 (dd.py:466): with a frozen bias the energy-path weight gradient of a Linear used to fall through to the
    immediate mm_tn while the force-path contributions to the same weight were deferred into the grouped launch --
    AccumulateGrad then added a real tensor to an unfilled buffer.  Every contribution to one leaf weight now takes the
    deferred path; the bias half alone is dropped.  One ForceTrainer step (open parameter-gradient batch) with all SchNet
    biases frozen, weight gradients against the oracle's autograd."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd.finetune_md17 import ForceTrainer
    from moleculesde_amd.synthetic import make_md17_batch
    kw = dict(hidden_channels=64, num_filters=32, num_interactions=3, num_gaussians=51, cutoff=10, readout="mean", node_class=119)
    torch.manual_seed(11)
    osch, ohead = R.SchNet(**kw), torch.nn.Linear(64, 1)
    sch, head = G.SchNet(**kw), torch.nn.Linear(64, 1)
    sch.load_state_dict(osch.state_dict()); head.load_state_dict(ohead.state_dict())
    for m in (osch, sch):
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.requires_grad_(False)
    cpu_b = make_md17_batch(2, seed=3, n_atoms=21)
    y, f_t = torch.randn(2), torch.randn(cpu_b.x.size(0), 3)
    pos = cpu_b.positions.clone().requires_grad_(True)
    e = ohead(osch(cpu_b.x, pos, cpu_b.batch)).squeeze(1)
    f = -torch.autograd.grad(e, pos, grad_outputs=torch.ones_like(e), create_graph=True, retain_graph=True)[0]
    lo = 0.05 * torch.nn.functional.l1_loss(e, y) + 0.95 * torch.nn.functional.l1_loss(f, f_t)
    lo.backward()
    ft = ForceTrainer(sch.to(dev), head.to(dev), lr=0.0)          # lr 0: the step leaves the parameters (and .grad) in place
    b = G.prepare_batch(cpu_b.clone(), dev)
    l = ft.step(b, y.to(dev), f_t.to(dev))
    assert_close(l, lo.detach(), 1e-4, 1e-6, "loss")
    frozen = [n for n, p in sch.named_parameters() if not p.requires_grad]
    assert frozen and all(p.grad is None for n, p in sch.named_parameters() if not p.requires_grad)
    gs = {n: p.grad for n, p in osch.named_parameters() if p.grad is not None}
    assert gs
    _grads_close(sch, gs, 2e-3, 2e-4, "SchNet double-backward grads, frozen biases")


def test_md17_force_path_runs_on_library_kernels(dev):
    """§8 a18: energy -> forces (create_graph) -> backward through the forces launches the kernels of the closed
    twice-differentiable operator set (moleculesde_amd.dd) and no torch operator inside SchNet.  What remains on torch is
    outside the library's boundary: the caller's own head / loss (finetune_MD17.py:50,60-72) and the sums the autograd
    ENGINE forms where one tensor feeds two operators (plain elementwise adds)."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import dd
    from moleculesde_amd.synthetic import make_md17_batch
    from torch.profiler import profile, ProfilerActivity
    torch.manual_seed(5)
    sch = G.SchNet(hidden_channels=64, num_filters=32, num_interactions=2, num_gaussians=51, cutoff=10, readout="mean",
                   node_class=119).to(dev)
    b = G.prepare_batch(make_md17_batch(4, seed=3, n_atoms=21), dev)

    def run():
        pos = b.positions.clone().requires_grad_(True)
        rep = sch(b.x, pos, b.batch)                           # [B, 64]; a linear head would be the caller's operator
        force = torch.autograd.grad(rep, pos, grad_outputs=torch.ones_like(rep), create_graph=True)[0]
        torch.autograd.backward([rep, force], [torch.ones_like(rep), torch.ones_like(force)])
    run()
    torch.cuda.synchronize()
    dd.CALLS.clear()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        run()
        torch.cuda.synchronize()
    assert dd.CALLS.get("msde_gemm_ex", 0) > 20 and dd.CALLS.get("msde_linear_bwd_w", 0) > 10, dd.CALLS
    assert dd.CALLS.get("msde_dd_rbf", 0) >= 3 and dd.CALLS.get("msde_dd_edge_scatter", 0) >= 1, dd.CALLS
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    ours = ("dd_", "gemm_ex", "gemm_small", "gemm_f32_mfma", "wgrad", "reduce_slabs", "colsum", "cfconv_aggregate", "radius_", "scan",
            "embedding_sum", "segment_sum", "transpose")
    foreign = sorted({n for n in names if not any(k in n for k in ours)})
    # allowed: the test's own clone / ones_like, and the autograd engine's gradient accumulation (a + b)
    allowed = ("FillFunctor", "Memcpy", "Memset", "CUDAFunctor_add", "direct_copy", "elementwise_kernel_manual_unroll")
    bad = [n for n in foreign if not any(k in n for k in allowed)]
    assert not bad, bad
    assert not any("Cijk" in n for n in names)          # no vendor GEMM


def test_fusion_sets_alias_flat_parameters(dev):
    """The trainer's optimiser lays every fusion set (embedding tables of one encoder; query/key/value/skip
    projections of one TransformerConv) out back to back, so hip.cat_params returns a VIEW of the parameters
    (no concatenation kernel), the view tracks optimiser updates, gradients still reach every member, and
    state-dict keys are untouched."""
    from moleculesde_amd import hip, pretrain
    args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=0)
    tr = pretrain.Trainer(args, dev)
    n_sets = 0
    for model in tr.models.values():
        for mod in model.modules():
            if not hasattr(mod, "fusion_sets"):
                continue
            for group in mod.fusion_sets():
                n_sets += 1
                cat = hip.cat_params(group)
                assert cat.data_ptr() == group[0].data_ptr(), "fusion set not contiguous in the flat buffer"
                assert cat.data_ptr() % 16 == 0
                assert torch.equal(cat, torch.cat([p.detach() for p in group], 0))
                (cat * 2.0).sum().backward()
                for p in group:
                    assert p.grad is not None and bool((p.grad == 2.0).all())
                    p.grad = None
    assert n_sets >= 6 + 2 * 4          # 6 embedding encoders (atom + 5 bond), 4 TransformerConv x (weights, biases)
    keys = set(tr.models["SDE_2Dto3D_model"].state_dict().keys())
    assert "score_network.gnn_layers.0.0.MHA.lin_query.weight" in keys and "edge_2D_emb.0.weight" in keys
    assert not any(k.startswith("_zero_bias") for k in keys)


# ------------------------------------------------- reference-generated 20-step curve through the HIP trainer ---
def test_golden_losscurve_through_hip_trainer(dev):
    """tests/golden/losscurve.npz was produced by the REFERENCE's own model files (oracle/make_golden.py): 20 Adam
    steps, bs 8, all three losses, dropout off, noise from torch.manual_seed(seed_base + step).  The product
    Trainer (HIP kernels, flat HIP Adam) replays it with CpuReplayNoise: every loss term of every step within the
    north star's 1e-3 relative -- flat, no noise-floor allowance."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import pretrain
    g = load_golden("losscurve.npz")
    args = pretrain.readme_args(emb_dim=32, num_layer=3, SchNet_num_filters=32, SchNet_num_interactions=2, lr=1e-3,
                                gnn_2d_lr_scale=1.0, gnn_3d_lr_scale=0.1)
    assert [args.lr * s for s in (1.0, 0.1, 1.0, 0.1)] == pytest.approx(list(g["lrs"]))
    torch.manual_seed(0)
    tr = pretrain.Trainer(args, dev)
    for k, m in tr.models.items():
        m.load_state_dict(sub(g, k + ".sd."))
        disable_dropout(m)
    b = G.prepare_batch(batch_from(g), dev)
    curve = []
    for step in range(20):
        tr.noise = G.CpuReplayNoise(int(g["seed_base"]) + step)
        tr.models["SDE_2Dto3D_model"].noise = tr.noise
        tr.models["SDE_3Dto2D_model"].noise = tr.noise
        loss, parts = tr.step(b)
        curve.append([float(loss), float(parts["CL"]), float(parts["2Dto3D"]), float(parts["3Dto2D"])])
    curve = np.array(curve)
    rel = np.abs(curve - g["curve"]) / np.abs(g["curve"])
    print("reference-generated loss curve through the HIP trainer: max rel err per column", rel.max(axis=0))
    assert rel.max() < 1e-3, (rel.max(axis=0), rel.argmax())


def test_checkpoint_round_trip_reference_layout(dev, tmp_path):
    """pretrain_MoleculeSDE.py:78-88: `model_complete.pth` = {'model_2D','model_3D','SDE_2Dto3D_model',
    'SDE_3Dto2D_model'} state dicts with the reference's keys.  Save from a trained-for-two-steps HIP Trainer, load
    into (a) the oracle classes (= the reference's key layout, strict) and (b) a fresh Trainer: identical tensors,
    identical next-step loss; finetune-style partial load (finetune_MD17.py:145-164 reads 'model_3D') works."""
    import json
    import os
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import pretrain
    from moleculesde_amd.synthetic import make_batch
    args = pretrain.readme_args(emb_dim=64)
    torch.manual_seed(6)
    tr = pretrain.Trainer(args, dev)
    b = G.prepare_batch(make_batch(16, seed=21), dev)
    for _ in range(2):
        tr.step(b)
    path = str(tmp_path / "model_complete.pth")
    tr.save(path)
    ck = torch.load(path, map_location="cpu")
    assert list(ck.keys()) == ["model_2D", "model_3D", "SDE_2Dto3D_model", "SDE_3Dto2D_model"]
    om = R.build_models(emb_dim=64, use_3d2d=True)
    for k in ck:
        om[k].load_state_dict(ck[k], strict=True)              # reference key layout, every key, every shape
    inv = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "inventory.json")))
    for k in ck:                                                 # same key ORDER as the reference at any width
        assert list(ck[k].keys()) == list(inv[k]["state_dict"].keys()), k
    # BatchNorm step counters were folded into the checkpoint (2 training-mode forwards)
    assert int(ck["model_2D"]["batch_norms.0.num_batches_tracked"]) == 2
    torch.manual_seed(7)
    tr2 = pretrain.Trainer(args, dev)
    for k in ck:
        tr2.models[k].load_state_dict(ck[k], strict=True)
    for k in ck:
        for n, v in tr.models[k].state_dict().items():
            assert torch.equal(v.cpu(), tr2.models[k].state_dict()[n].cpu()), (k, n)
    for t in (tr, tr2):
        for m in t.models.values():
            disable_dropout(m)
        t.noise = G.CpuReplayNoise(9)
        t.models["SDE_2Dto3D_model"].noise = t.noise
        t.models["SDE_3Dto2D_model"].noise = t.noise
    l1, _ = tr.losses(b)
    l2, _ = tr2.losses(b)
    assert_close(l2, l1.detach(), 1e-6, 0, "loss after reload")
    sch = G.SchNet(hidden_channels=64, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=10, readout="mean",
                   node_class=119)
    sch.load_state_dict(ck["model_3D"])                         # finetune_MD17.py:151


# ------------------------------------------------------------------ fused dense-head kernels (a12-a14) ---
@pytest.mark.parametrize("emb,bs,reduce_mean,anneal", [(300, 64, True, 0.0), (32, 7, False, 1.5)])
def test_fused_dense_head_vs_operator_path(dev, emb, bs, reduce_mean, anneal):
    """csrc/dense_head.hip + gemm_ex (one autograd node, ragged, no torch operator) against the operator-by-operator
    path of the same module (itself pinned to the genuine-reference golden): both losses, the gradient of the 3D
    representation and every parameter gradient, same replayed noise."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd.geom3d import sde_3d_to_2d as S
    from moleculesde_amd.synthetic import make_batch
    torch.manual_seed(12)
    m = _s32(G, emb).to(dev).train()
    b = G.prepare_batch(make_batch(bs, seed=41), dev)
    h3 = torch.randn(b.x.size(0), emb, device=dev)
    from moleculesde_amd.geom3d import dense_head as DH
    res = {}
    for fused in (False, True):
        S.USE_FUSED_HEAD = fused
        calls0 = DH.FUSED_CALLS
        try:
            m.noise = G.CpuReplayNoise(99)
            h = h3.clone().requires_grad_(True)
            for p in m.parameters():
                p.grad = None
            lx, la = m(h, b, reduce_mean=reduce_mean, continuous=True, train=True, anneal_power=anneal)
            assert DH.FUSED_CALLS == calls0 + (1 if fused else 0), "wrong path ran"
            (lx * 0.7 + la * 1.3).backward()
            res[fused] = (lx.detach(), la.detach(), h.grad.clone(), {n: (p.grad.clone() if p.grad is not None else None)
                                                                    for n, p in m.named_parameters()})
        finally:
            S.USE_FUSED_HEAD = True
    (lx0, la0, gh0, gp0), (lx1, la1, gh1, gp1) = res[False], res[True]
    assert_close(lx1, lx0, 2e-5, 0, "loss_x")
    assert_close(la1, la0, 2e-5, 0, "loss_adj")
    assert_close(gh1, gh0, 1e-3, 1e-5 * float(gh0.abs().max()), "grad h3")
    scale = max(float(v.abs().max()) for v in gp0.values() if v is not None)
    for n, v0 in gp0.items():
        v1 = gp1[n]
        if v0 is None or float(v0.abs().max()) == 0.0:
            assert v1 is None or float(v1.abs().max()) <= 1e-6 * scale, n      # unused output branch of the last layer
            continue
        assert v1 is not None, n
        assert_close(v1, v0, 2e-3, 2e-5 * scale, f"grad {n}")


def test_fused_dense_head_device_noise_statistics(dev):
    """The in-kernel noise path (no replay): draws are N(0,1)-like, symmetric, masked; two calls differ; the losses are
    finite and of the size the replayed path gives."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd.synthetic import make_batch
    torch.manual_seed(13)
    m = _s32(G, 64).to(dev).train()
    b = G.prepare_batch(make_batch(128, seed=42), dev)
    h3 = torch.randn(b.x.size(0), 64, device=dev)
    m.noise = G.DeviceNoise(seed=5)
    l1 = m(h3, b, reduce_mean=True, continuous=True, train=True, anneal_power=0)
    l2 = m(h3, b, reduce_mean=True, continuous=True, train=True, anneal_power=0)
    m.noise = G.CpuReplayNoise(3)
    l3 = m(h3, b, reduce_mean=True, continuous=True, train=True, anneal_power=0)
    for a in (l1, l2, l3):
        assert all(bool(torch.isfinite(t)) for t in a)
    assert float(l1[0]) != float(l2[0]) and float(l1[1]) != float(l2[1])
    assert abs(float(l1[0]) / float(l3[0]) - 1) < 0.2 and abs(float(l1[1]) / float(l3[1]) - 1) < 0.3


def _set_parameters_like_generator(net, base):
    """Same deterministic fill as oracle/make_golden_dense_prod.py::set_parameters."""
    with torch.no_grad():
        for i, (_, p) in enumerate(net.named_parameters()):
            gen = torch.Generator().manual_seed(base + i)
            if p.dim() == 1:
                p.copy_(torch.rand(p.shape, generator=gen) * 0.6 - 0.3)
            else:
                fan = p.shape[1] if p.shape[0] != p.shape[1] else p.shape[0]
                p.copy_(torch.randn(p.shape, generator=gen) / fan ** 0.5)


@pytest.mark.parametrize("fused", [True, False])
def test_golden_dense_head_production_dims_genuine(dev, fused):
    """tests/golden/dense_head_prod.npz: the GENUINE reference classes (no stand-ins) at the head configuration of
    pretrain_MoleculeSDE.py:310-315 on a ragged batch incl. a bond-less atom.  The fused kernels
    (geom3d.dense_head.score_networks) and the operator path both reproduce scores, the node-feature gradient and
    every parameter gradient."""
    from moleculesde_amd.geom3d import sde_3d_to_2d as S
    from moleculesde_amd.geom3d import dense_head as DH
    g = load_golden("dense_head_prod.npz")
    Fd, nout = g["x"].shape[-1], g["score_node"].shape[-1]
    edge = S.EdgeScoreNetwork_dense(dim3D=Fd, nhid=16, num_layers=3, num_linears=3, c_init=2, c_hid=8, c_final=4, adim=16,
                                    num_heads=4, conv="MLP")
    node = S.NodeScoreNetwork_dense(nfeat=Fd, depth=4, nhid=16, nout=nout)
    assert [k for k, _ in edge.named_parameters()] == list(g["param_names_edge"])
    assert [k for k, _ in node.named_parameters()] == list(g["param_names_node"])
    _set_parameters_like_generator(edge, 7000)
    _set_parameters_like_generator(node, 9000)
    edge.to(dev); node.to(dev)
    x = torch.from_numpy(g["x"]).to(dev).requires_grad_(True)
    a = torch.from_numpy(g["adj"]).to(dev)
    flags = torch.from_numpy(g["flags"]).to(dev)
    if fused:
        assert DH.fused_supported(edge, node, x.size(1))
        se, sn = DH.score_networks(edge, node, x, a, flags)
    else:
        se, sn = edge(x, a, flags), node(x, a, flags)
    assert_close(se, g["score_edge"], 1e-4, 1e-5, "edge score")
    assert_close(sn, g["score_node"], 1e-4, 1e-5, "node score")
    (se.pow(2).sum() + sn.pow(2).sum()).backward()
    assert_close(x.grad, g["grad_x"], 1e-3, 1e-4 * float(np.abs(g["grad_x"]).max()), "grad x")
    _grads_close(edge, sub(g, "edge.grad."), 1e-3, 1e-4, "edge net")
    _grads_close(node, sub(g, "node.grad."), 1e-3, 1e-4, "node net")


@pytest.mark.parametrize("variant", ["", "_02"])
def test_dense_head_launches_no_torch_operator(dev, variant):
    """zero ATen launches inside the head (r3: also for the concatenating variant _02).  Every kernel of a fused forward + backward (device
    noise) carries a name from libmsde_hip.so; nothing from at::native / rocBLAS / hipBLASLt / rocclr copy-fill."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import hip, pretrain
    from moleculesde_amd.synthetic import make_batch
    from torch.profiler import profile, ProfilerActivity
    torch.manual_seed(3)
    args = pretrain.readme_args(emb_dim=64, SDE_3Dto2D_model="SDEModel3Dto2D_node_adj_dense" + variant)
    tr = pretrain.Trainer(args, dev)                      # FlatAdam lays the stacked parameters out back to back
    m = tr.models["SDE_3Dto2D_model"]
    m.noise = G.DeviceNoise(seed=11)
    b = G.prepare_batch(make_batch(64, seed=43), dev)
    h3 = torch.randn(b.x.size(0), 64, device=dev, requires_grad=True)

    def run():
        tr.opt.zero_grad()
        h3.grad = None
        lx, la = m(h3, b, reduce_mean=True, continuous=True, train=True, anneal_power=0)
        out = torch.stack([lx, la])
        slabs.begin_param_grad_batch(tr.opt.params)
        try:
            torch.autograd.backward(out, torch.ones_like(out))
        finally:
            slabs.finish_param_grad_batch()
    run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        run()
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    assert len(names) > 30, names
    ours = ("dense_", "gemm_ex", "gemm_small", "gemm_rsa", "gemm_t2", "transpose_", "gemm_grouped_wgrad", "reduce_slabs", "colsum", "bn_")
    foreign = [n for n in names if not any(k in n for k in ours)]
    # the two stack/ones_like glue ops of THIS TEST (outside the head) are the only operator kernels allowed
    foreign = [n for n in foreign if "CatArrayBatchedCopy" not in n and "FillFunctor" not in n and "Memcpy" not in n
               and "Memset" not in n]
    assert not foreign, sorted(set(foreign))


# ------------------------------------------------------------------ §8 f3: _01 and the VP SDE on the HIP path ---
@pytest.mark.parametrize("tag", ["m01", "vp"])
def test_golden_f3_variants_2d3d(dev, tag):
    """SDEModel2Dto3D_01 (VE) and SDEModel2Dto3D_02 with the VP SDE on the HIP kernels against the fixture produced by
    the reference's own files (tests/golden/f3_variants.npz): loss, gradients, get_score, VP predictor steps."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import sampler
    g = load_golden("f3_variants.npz")
    b = G.prepare_batch(batch_from(g), dev)
    cls, sde_type = (G.SDEModel2Dto3D_01, "VE") if tag == "m01" else (G.SDEModel2Dto3D_02, "VP")
    m = disable_dropout(cls(emb_dim=16, hidden_dim=32, beta_schedule=None, beta_min=0.2, beta_max=1.0,
                            num_diffusion_timesteps=1000, SDE_type=sde_type, use_extend_graph=True))
    assert list(m.state_dict().keys()) == list(sub(g, f"{tag}.sd.").keys())
    m.load_state_dict(sub(g, f"{tag}.sd."))
    m.to(dev).train()
    m.noise = G.CpuReplayNoise(int(g[f"{tag}.seed"]))
    h2 = torch.from_numpy(g["h2"]).to(dev).requires_grad_(True)
    loss = m(h2, b, anneal_power=0)["position"]
    assert_close(loss, g[f"{tag}.loss"], 1e-4, 1e-6, "loss")
    loss.backward()
    assert_close(h2.grad, g[f"{tag}.grad_h2"], 1e-3, 1e-4 * float(np.abs(g[f"{tag}.grad_h2"]).max()), "grad h2")
    _grads_close(m, sub(g, f"{tag}.grad."), 1e-3, 2e-4, tag)
    m.eval()
    pos, t = torch.from_numpy(g[f"{tag}.score_pos"]).to(dev), torch.from_numpy(g[f"{tag}.score_t"]).to(dev)
    sc = m.get_score(h2.detach(), b, pos, None, t)
    assert_close(sc, g[f"{tag}.score"], 1e-3, 1e-4 * float(np.abs(g[f"{tag}.score"]).max()), "get_score")
    if tag == "vp":
        x = pos.clone()
        for i, tv in enumerate(g["vp.pred_ts"]):
            vt = torch.full((x.size(0),), float(tv), device=dev)
            x, _ = sampler.predictor_update(m.sde_pos, m, h2.detach(), b, x, vt,
                                            noise=torch.from_numpy(g["vp.pred_noise"][i]).to(dev))
            assert_close(x, g["vp.pred_traj"][i], 1e-3, 1e-4, f"VP predictor step {i}")


# ------------------------------------------------------------------ robustness of the step machinery ---
def _quiet_trainer(dev, emb=64, seed=5):
    from moleculesde_amd import pretrain
    args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=0, emb_dim=emb)
    torch.manual_seed(seed)
    tr = pretrain.Trainer(args, dev)
    tr.overlap_streams = False
    return tr


def test_eager_steps_with_gpu_far_behind_host(dev):
    """The eager path uploads pointer tables (Adam chunk table, slab rows, grouped-GEMM problems) from pinned host
    images.  With the GPU several steps behind the host (a long sleep kernel in front), the images are rotated / guarded
    by events, so the result equals a run that synchronises after every step."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd.synthetic import make_batch
    batches = [G.prepare_batch(make_batch(12, seed=70 + s, sizes=[5 + (s + i) % 9 for i in range(12)]), dev) for s in range(4)]
    res = []
    for delayed in (False, True):
        tr = _quiet_trainer(dev)
        for m in tr.models.values():
            disable_dropout(m)
        tr.noise = G.CpuReplayNoise(123)
        tr.models["SDE_2Dto3D_model"].noise = tr.noise
        if delayed:
            torch.cuda._sleep(int(2.0e9))              # ~1 s of GPU time queued in front of everything
        for s in range(6):
            tr.step(batches[s % 4])                    # different shapes -> different gradient / slab addresses per step
            if not delayed:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        res.append(tr.opt.flat_p.clone())
    assert torch.equal(res[0], res[1])


def test_replay_after_workspaces_grew(dev):
    """capture on a small batch, eager step on a LARGER one (grow-on-demand workspaces are replaced), replay the small
    graph: outgrown workspaces stay alive for the graph, so the replay matches an eager step."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd.synthetic import make_batch
    small = G.prepare_batch(make_batch(8, seed=81), dev)
    big = G.prepare_batch(make_batch(96, seed=82), dev)

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
    outs = []
    for use_graph in (True, False):
        tr = _quiet_trainer(dev, seed=6)
        for m in tr.models.values():
            disable_dropout(m)
        tr.noise = Fixed()
        tr.models["SDE_2Dto3D_model"].noise = tr.noise
        tr.step(small)
        if use_graph:
            tr.capture(small)
        tr.step(big)                                   # larger shape: scratch / slab arena / BN workspace grow
        for _ in range(2):
            (tr.step_graph if use_graph else tr.step)(small)
        torch.cuda.synchronize()
        outs.append(tr.opt.flat_p.clone())
    d = float((outs[0] - outs[1]).norm() / outs[1].norm())
    assert d < 1e-5, d


def test_eager_steps_after_capture_draw_fresh_dropout_masks(dev):
    """After a capture the dropout seeds come from the device step counter; the eager step advances it too, so two
    eager steps on the same batch with identical other noise still differ through the attention / FFN dropout."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd.synthetic import make_batch
    tr = _quiet_trainer(dev, seed=7)
    b = G.prepare_batch(make_batch(16, seed=83), dev)
    tr.step(b)
    tr.capture(b)

    class Fixed(G.DeviceNoise):
        z, t = torch.randn(4096, 3).to(dev), torch.randint(0, 1000, (64,)).to(dev)
        def randn_like(self, x): return self.z[:x.size(0)].clone()
        def randint(self, high, size, device): return self.t[:size[0]].clone()
        def randperm_pair(self, n, device):
            p = torch.arange(n - 1, -1, -1, dtype=torch.int32, device=device)
            return p, p
    tr.noise = Fixed()
    tr.models["SDE_2Dto3D_model"].noise = tr.noise
    sd = {k: {n: v.clone() for n, v in m.state_dict().items()} for k, m in tr.models.items()}
    losses = []
    for _ in range(2):
        for k, m in tr.models.items():
            m.load_state_dict(sd[k])
        _, parts = tr.step(b)
        losses.append(float(parts["2Dto3D"]))
    assert losses[0] != losses[1], losses
