"""oracle/restate.py vs the golden vectors produced by the reference's own files
(oracle/make_golden.py).  Runs everywhere (CPU); pins the oracle before it is trusted as the
checker for the HIP path."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import restate as R
from helpers import GOLDEN, assert_close, batch_from, disable_dropout, grads_close, load_golden, sub

TOY = dict(emb=16, filters=16, interactions=2, gaussians=51)


def _s23(E):
    return disable_dropout(R.SDEModel2Dto3D_02(emb_dim=E, hidden_dim=32, beta_min=0.2, beta_max=1.0,
                                               num_diffusion_timesteps=1000, beta_schedule=None, SDE_type="VE",
                                               use_extend_graph=True))


def _s32(E):
    return R.SDEModel3Dto2D_node_adj_dense(dim3D=E, c_init=2, c_hid=8, c_final=4, num_heads=4, adim=16, nhid=16,
                                           num_layers=4, emb_dim=E, num_linears=3, beta_min=0.1, beta_max=1.0,
                                           num_diffusion_timesteps=1000, SDE_type="VE", num_class_X=119,
                                           noise_on_one_hot=True)


def test_dense_head_genuine():
    g = load_golden("dense_head.npz")
    edge = R.EdgeScoreNetwork_dense(dim3D=12, nhid=8, num_layers=3, num_linears=3, c_init=2, c_hid=4, c_final=2,
                                    adim=8, num_heads=4, conv="MLP")
    node = R.NodeScoreNetwork_dense(nfeat=12, depth=3, nhid=8, nout=7)
    edge.load_state_dict(sub(g, "edge.sd."))
    node.load_state_dict(sub(g, "node.sd."))
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    a = torch.from_numpy(g["adj"]).requires_grad_(True)
    flags = torch.from_numpy(g["flags"])
    se, sn = edge(x, a, flags), node(x, a, flags)
    assert_close(se, g["score_edge"], 1e-5, 1e-6, "edge score")
    assert_close(sn, g["score_node"], 1e-5, 1e-6, "node score")
    (se.pow(2).sum() + sn.pow(2).sum()).backward()
    assert_close(x.grad, g["grad_x"], 1e-4, 1e-5, "grad x")
    assert_close(a.grad, g["grad_adj"], 1e-4, 1e-5, "grad adj")
    grads_close(edge, {k: v for k, v in sub(g, "edge.grad.").items()}, 1e-4, 1e-5, "edge")
    grads_close(node, {k: v for k, v in sub(g, "node.grad.").items()}, 1e-4, 1e-5, "node")


def test_vesde_golden():
    g = load_golden("vesde.npz")
    t = torch.from_numpy(g["t"])
    ve = R.VESDE(0.2, 1.0, 1000)
    _, std = ve.marGINal_prob(torch.zeros(5, 3), t)
    _, G = ve.discretize(torch.zeros(5, 3), t)
    _, diff = ve.sde(torch.zeros(5, 3), t)
    assert_close(std, g["std"], 1e-6, 0, "std")
    assert_close(G, g["G"], 1e-6, 1e-9, "G")
    assert_close(diff, g["diffusion"], 1e-6, 0, "diffusion")
    assert_close(ve.discrete_sigmas, g["discrete_sigmas"], 1e-6, 0, "sigmas")
    _, sd = R.VESDE(0.1, 1.0, 1000).marGINal_prob(torch.zeros(5, 2, 2), t)
    assert_close(sd, g["std_dense"], 1e-6, 0, "std dense")


def test_toy_gnn():
    g = load_golden("toy_gnn.npz")
    b = batch_from(g)
    m = R.GNN(3, TOY["emb"], JK="last", drop_ratio=0, gnn_type="GIN")
    m.load_state_dict(sub(g, "sd."))
    out = m(b.x, b.edge_index, b.edge_attr)
    assert_close(out, g["out"], 1e-5, 1e-6, "gnn out")
    out.pow(2).sum().backward()
    grads_close(m, sub(g, "grad."), 1e-4, 1e-5, "gnn")
    # BatchNorm running statistics after one training-mode forward
    for k, v in sub(g, "sd_after.").items():
        assert_close(m.state_dict()[k], v, 1e-5, 1e-6, "sd_after." + k)


def test_toy_schnet():
    g = load_golden("toy_schnet.npz")
    b = batch_from(g)
    m = R.SchNet(hidden_channels=TOY["emb"], num_filters=TOY["filters"], num_interactions=TOY["interactions"],
                 num_gaussians=TOY["gaussians"], cutoff=10, readout="mean", node_class=119)
    m.load_state_dict(sub(g, "sd."))
    pos = b.positions.clone().requires_grad_(True)
    out, h = m(b.x[:, 0], pos, b.batch, return_latent=True)
    assert_close(out, g["out"], 1e-5, 1e-6, "schnet out")
    assert_close(h, g["h"], 1e-5, 1e-6, "schnet h")
    (h.pow(2).sum() + out.sum()).backward()
    assert_close(pos.grad, g["grad_pos"], 1e-4, 1e-5, "grad pos")
    grads_close(m, sub(g, "grad."), 1e-4, 1e-5, "schnet")


def test_toy_sde2d3d():
    g = load_golden("toy_sde2d3d.npz")
    b = batch_from(g)
    m = _s23(TOY["emb"])
    m.load_state_dict(sub(g, "sd."))
    h2 = torch.from_numpy(g["h2"]).requires_grad_(True)
    torch.manual_seed(int(g["seed"]))
    loss = m(h2, b, anneal_power=0)["position"]
    assert_close(loss, g["loss"], 1e-5, 1e-6, "loss 2d3d")
    loss.backward()
    assert_close(h2.grad, g["grad_h2"], 1e-4, 1e-6, "grad h2")
    grads_close(m, sub(g, "grad."), 1e-4, 1e-5, "sde2d3d")
    for k, v in sub(g, "sd_after.").items():
        assert_close(m.state_dict()[k], v, 1e-5, 1e-6, "sd_after." + k)
    m.eval()
    score = m.get_score(torch.from_numpy(g["h2"]), b, torch.from_numpy(g["gs_pos"]), None, torch.from_numpy(g["gs_t_pos"]))
    assert_close(score, g["gs_score"], 1e-4, 1e-5, "get_score")


def test_toy_sde3d2d():
    g = load_golden("toy_sde3d2d.npz")
    b = batch_from(g)
    m = _s32(TOY["emb"])
    m.load_state_dict(sub(g, "sd."))
    h3 = torch.from_numpy(g["h3"]).requires_grad_(True)
    torch.manual_seed(int(g["seed"]))
    lx, la = m(h3, b, reduce_mean=True, continuous=True, train=True, anneal_power=0)
    assert_close(lx, g["loss_x"], 1e-5, 1e-6, "loss_x")
    assert_close(la, g["loss_adj"], 1e-5, 1e-6, "loss_adj")
    (lx + la).backward()
    assert_close(h3.grad, g["grad_h3"], 1e-4, 1e-6, "grad h3")
    grads_close(m, sub(g, "grad."), 1e-4, 1e-5, "sde3d2d")


def test_qm9_schnet_config1():
    """BASELINE.json configs[0]: QM9-shaped SchNet forward, batch 32, CPU."""
    g = load_golden("qm9_schnet.npz")
    b = batch_from(g)
    assert b.x.dim() == 1 and b.num_graphs == 32
    m = R.SchNet(hidden_channels=32, num_filters=32, num_interactions=3, num_gaussians=51, cutoff=10,
                 readout="mean", node_class=119)
    m.load_state_dict(sub(g, "sd."))
    out, h = m(b.x, b.positions, b.batch, return_latent=True)
    assert_close(out, g["out"], 1e-5, 1e-6, "qm9 out")
    assert_close(h, g["h"], 1e-5, 1e-6, "qm9 h")


def test_losscurve_20_steps():
    """20 Adam steps, bs 8, all three losses: loss curve within 1e-3 relative (BASELINE target)."""
    g = load_golden("losscurve.npz")
    lb = batch_from(g)
    models = {
        "model_2D": R.GNN(3, 32, JK="last", drop_ratio=0, gnn_type="GIN"),
        "model_3D": R.SchNet(hidden_channels=32, num_filters=32, num_interactions=2, num_gaussians=51, cutoff=10,
                             readout="mean", node_class=119),
        "SDE_2Dto3D_model": _s23(32),
        "SDE_3Dto2D_model": _s32(32),
    }
    for k, m in models.items():
        m.load_state_dict(sub(g, k + ".sd."))
    lrs = g["lrs"]
    opt = torch.optim.Adam([{"params": m.parameters(), "lr": float(lr)} for m, lr in zip(models.values(), lrs)],
                           lr=1e-3, weight_decay=0.0)
    curve = []
    for step in range(20):
        torch.manual_seed(int(g["seed_base"]) + step)
        loss, parts = R.pretrain_losses(models, lb.clone(), T=0.1)
        opt.zero_grad()
        loss.backward()
        opt.step()
        curve.append([loss.item(), parts["CL"].item(), parts["2Dto3D"].item(), parts["3Dto2D"].item()])
    curve = np.array(curve)
    rel = np.abs(curve - g["curve"]) / np.abs(g["curve"])
    assert rel.max() < 1e-3, rel.max()


def test_sampler_predictor_corrector():
    g = load_golden("sampler.npz")
    b = batch_from(g)
    m = _s23(TOY["emb"])
    m.load_state_dict(sub(g, "sd."))
    m.eval()
    rep = torch.from_numpy(g["rep"])
    pos = torch.from_numpy(g["pos0"])
    n = pos.size(0)
    for i, tval in enumerate(torch.from_numpy(g["ts"])):
        vec_t = torch.ones(n) * tval
        pos, _ = R.corrector_update(m.sde_pos, m, rep, b, pos, vec_t, float(g["snr"]), float(g["scale_eps"]), 1,
                                    noises=[torch.from_numpy(g["noise_corr"][i])])
        pos, _ = R.predictor_update(m.sde_pos, m, rep, b, pos, vec_t, noise=torch.from_numpy(g["noise_pred"][i]))
        assert_close(pos, g["traj"][i], 1e-4, 1e-5, f"sampler step {i}")


def test_inventory_readme_configuration():
    """Parameter counts and state_dict keys/shapes at the README configuration."""
    with open(os.path.join(GOLDEN, "inventory.json")) as f:
        inv = json.load(f)
    models = R.build_models()
    total = 0
    for k, m in models.items():
        sd = m.state_dict()
        assert list(sd.keys()) == list(inv[k]["state_dict"].keys()), k
        for n, (shape, dtype) in inv[k]["state_dict"].items():
            assert list(sd[n].shape) == shape and str(sd[n].dtype) == dtype, (k, n)
        tr = sum(p.numel() for p in m.parameters() if p.requires_grad)
        assert tr == inv[k]["trainable"], k
        total += tr
    assert total == 4668227


@pytest.mark.parametrize("tag", ["m01", "vp"])
def test_f3_variants_2d3d(tag):
    """SDEModel2Dto3D_01 (VE) and SDEModel2Dto3D_02 with the VP SDE, fixture from the reference's own
    files (oracle/make_golden_f3.py): loss, gradients, get_score, and (VP) three reverse-diffusion predictor steps."""
    g = load_golden("f3_variants.npz")
    b = batch_from(g)
    cls, sde_type = (R.SDEModel2Dto3D_01, "VE") if tag == "m01" else (R.SDEModel2Dto3D_02, "VP")
    m = disable_dropout(cls(emb_dim=16, hidden_dim=32, beta_schedule=None, beta_min=0.2, beta_max=1.0,
                            num_diffusion_timesteps=1000, SDE_type=sde_type, use_extend_graph=True))
    m.load_state_dict(sub(g, f"{tag}.sd."))
    h2 = torch.from_numpy(g["h2"]).requires_grad_(True)
    torch.manual_seed(int(g[f"{tag}.seed"]))
    loss = m(h2, b.clone(), anneal_power=0)["position"]
    assert_close(loss, g[f"{tag}.loss"], 1e-5, 1e-6, "loss")
    loss.backward()
    assert_close(h2.grad, g[f"{tag}.grad_h2"], 1e-4, 1e-6, "grad h2")
    grads_close(m, sub(g, f"{tag}.grad."), 1e-4, 1e-5, tag)
    m.eval()
    pos, t = torch.from_numpy(g[f"{tag}.score_pos"]), torch.from_numpy(g[f"{tag}.score_t"])
    assert_close(m.get_score(h2.detach(), b.clone(), pos, None, t), g[f"{tag}.score"], 1e-4, 1e-5, "get_score")
    if tag == "vp":
        x = pos.clone()
        for i, tv in enumerate(g["vp.pred_ts"]):
            vt = torch.full((x.size(0),), float(tv))
            f, G = m.sde_pos.reverse_discretize(m, x, h2.detach(), b.clone(), vt)
            x = (x - f) + G[:, None] * torch.from_numpy(g["vp.pred_noise"][i])
            assert_close(x, g["vp.pred_traj"][i], 1e-4, 1e-5, f"VP predictor step {i}")
