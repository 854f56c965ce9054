"""Generate tests/golden/*.npz by executing the reference's own files (this container only).

    python -m oracle.make_golden          # needs /root/reference; writes tests/golden/

Each fixture is DATA ONLY: a state_dict, seeded inputs, and the outputs/gradients the reference
produced.  No reference source is stored.  The parity tests replay the fixtures through
oracle/restate.py (CPU, everywhere) and through the HIP path (GPU box).

Fixtures:
  dense_head.npz      genuine import (no stand-ins): EdgeScoreNetwork_dense / NodeScoreNetwork_dense fwd + grads
  vesde.npz           genuine VESDE marGINal_prob / discretize known answers
  toy_gnn.npz         GNN (GIN) fwd + grads, 3 molecules, emb 16           (verbatim on stand-ins)
  toy_schnet.npz      SchNet fwd + grads, same toy
  toy_sde2d3d.npz     SDEModel2Dto3D_02 forward loss (seeded noise, dropout off) + get_score + grads
  toy_sde3d2d.npz     SDEModel3Dto2D_node_adj_dense forward losses (seeded noise) + grads
  qm9_schnet.npz      config 1: QM9-shaped bs 32 SchNet forward
  losscurve.npz       20 Adam steps, bs 8, dropout off, seeded noise: the loss values
  sampler.npz         5 predictor-corrector steps with the genuine VESDE.reverse().discretize
  inventory.json      parameter counts + state_dict keys/shapes at the README configuration
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from moleculesde_amd.synthetic import make_batch, make_qm9_batch  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def disable_dropout(model):
    """Dropout is hard-wired on in the 2D->3D score net (equivariant_scorenetwork.py:93,108); the
    parity fixtures switch it off on both sides."""
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
            m.dropout = 0.0
    return model


def sd_np(model, prefix="sd."):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def grads_np(model, prefix="grad."):
    return {prefix + k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters() if p.grad is not None}


def batch_np(b, prefix="batch."):
    d = {prefix + k: getattr(b, k).detach().cpu().numpy() for k in b.tensor_keys()}
    d[prefix + "num_graphs"] = np.int64(b.num_graphs)
    return d


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays")


TOY = dict(emb=16, filters=16, interactions=2, gaussians=51)


def build_ref(ns, E, filters, interactions, gaussians=51, layers=3):
    gnn = ns.GNN(layers, E, JK="last", drop_ratio=0, gnn_type="GIN")
    sch = ns.SchNet(hidden_channels=E, num_filters=filters, num_interactions=interactions, num_gaussians=gaussians,
                    cutoff=10, readout="mean", node_class=119)
    s23 = disable_dropout(ns.SDEModel2Dto3D_02(emb_dim=E, hidden_dim=32, beta_min=0.2, beta_max=1.0,
                                               num_diffusion_timesteps=1000, beta_schedule=None, SDE_type="VE",
                                               use_extend_graph=True))
    s32 = ns.SDEModel3Dto2D_node_adj_dense(dim3D=E, c_init=2, c_hid=8, c_final=4, num_heads=4, adim=16, nhid=16,
                                           num_layers=4, emb_dim=E, num_linears=3, beta_min=0.1, beta_max=1.0,
                                           num_diffusion_timesteps=1000, SDE_type="VE", num_class_X=119,
                                           noise_on_one_hot=True)
    return gnn, sch, s23, s32


def main():
    os.makedirs(OUT, exist_ok=True)
    g = ref_loader.genuine()
    ns = ref_loader.verbatim()

    # ---- (i) dense head, genuine import --------------------------------------------------
    torch.manual_seed(100)
    B, N, Fd = 3, 5, 12
    edge_net = g.EdgeScoreNetwork_dense(dim3D=Fd, nhid=8, num_layers=3, num_linears=3, c_init=2, c_hid=4, c_final=2,
                                        adim=8, num_heads=4, conv="MLP")
    node_net = g.NodeScoreNetwork_dense(nfeat=Fd, depth=3, nhid=8, nout=7)
    x = torch.randn(B, N, Fd, requires_grad=True)
    a = torch.randn(B, N, N)
    a = ((a + a.transpose(1, 2)) * 0.5)
    flags = torch.ones(B, N)
    flags[1, 4] = 0
    flags[2, 3:] = 0
    a = (a * flags[:, :, None] * flags[:, None, :]).requires_grad_(True)
    se = edge_net(x, a, flags)
    sn = node_net(x, a, flags)
    (se.pow(2).sum() + sn.pow(2).sum()).backward()
    save("dense_head.npz", x=x.detach().numpy(), adj=a.detach().numpy(), flags=flags.numpy(),
         score_edge=se.detach().numpy(), score_node=sn.detach().numpy(),
         grad_x=x.grad.numpy(), grad_adj=a.grad.numpy(),
         **sd_np(edge_net, "edge.sd."), **grads_np(edge_net, "edge.grad."),
         **sd_np(node_net, "node.sd."), **grads_np(node_net, "node.grad."))

    # ---- (ii) VESDE known answers ------------------------------------------------------
    ve = g.SDE_sparse.VESDE(sigma_min=0.2, sigma_max=1.0, N=1000)
    t = torch.tensor([1e-6, 0.25, 0.5, 0.999, 1.0])
    xx = torch.zeros(5, 3)
    _, std = ve.marGINal_prob(xx, t)
    _, G = ve.discretize(xx, t)
    _, diff = ve.sde(xx, t)
    ved = g.SDE_dense.VESDE(sigma_min=0.1, sigma_max=1.0, N=1000)
    _, stdd = ved.marGINal_prob(torch.zeros(5, 2, 2), t)
    save("vesde.npz", t=t.numpy(), std=std.numpy(), G=G.numpy(), diffusion=diff.numpy(), std_dense=stdd.numpy(),
         discrete_sigmas=ve.discrete_sigmas.numpy())

    # ---- (iii) toy goldens, verbatim on stand-ins --------------------------------------
    torch.manual_seed(7)
    toy = make_batch(3, seed=11, sizes=[5, 9, 7])
    gnn, sch, s23, s32 = build_ref(ns, TOY["emb"], TOY["filters"], TOY["interactions"], TOY["gaussians"])
    # perturb BN-affected zero-init params so gradients are informative
    for m in (gnn, sch, s23, s32):
        for p in m.parameters():
            if p.requires_grad and p.dim() == 1:
                p.data.add_(0.05 * torch.randn_like(p))

    sd0 = sd_np(gnn)
    h2 = gnn(toy.x, toy.edge_index, toy.edge_attr)
    h2.pow(2).sum().backward()
    save("toy_gnn.npz", out=h2.detach().numpy(), **batch_np(toy), **sd0, **grads_np(gnn),
         **sd_np(gnn, "sd_after."))

    pos = toy.positions.clone().requires_grad_(True)
    out3, h3 = sch(toy.x[:, 0], pos, toy.batch, return_latent=True)
    (h3.pow(2).sum() + out3.sum()).backward()
    save("toy_schnet.npz", out=out3.detach().numpy(), h=h3.detach().numpy(), grad_pos=pos.grad.numpy(),
         **batch_np(toy), **sd_np(sch), **grads_np(sch))

    gnn.zero_grad(); sch.zero_grad()
    h2d = h2.detach().clone().requires_grad_(True)
    tb = toy.clone()
    sd0 = sd_np(s23)
    torch.manual_seed(21)
    loss23 = s23(h2d, tb, anneal_power=0)["position"]
    loss23.backward()
    tb2 = toy.clone()
    torch.manual_seed(22)
    t_pos = torch.rand(toy.x.size(0)) * 0.9 + 0.05
    pos_pert = toy.positions + 0.3 * torch.randn_like(toy.positions)
    s23.eval()  # get_score is used at inference: BN in eval mode
    score = s23.get_score(h2.detach(), tb2, pos_pert, None, t_pos)
    s23.train()
    save("toy_sde2d3d.npz", seed=np.int64(21), h2=h2.detach().numpy(), loss=loss23.detach().numpy(),
         grad_h2=h2d.grad.numpy(), gs_t_pos=t_pos.numpy(), gs_pos=pos_pert.numpy(), gs_score=score.numpy(),
         **batch_np(toy), **sd0, **grads_np(s23), **sd_np(s23, "sd_after."))

    h3d = h3.detach().clone().requires_grad_(True)
    torch.manual_seed(31)
    lx, la = s32(h3d, toy.clone(), reduce_mean=True, continuous=True, train=True, anneal_power=0)
    (lx + la).backward()
    save("toy_sde3d2d.npz", seed=np.int64(31), h3=h3.detach().numpy(), loss_x=lx.detach().numpy(),
         loss_adj=la.detach().numpy(), grad_h3=h3d.grad.numpy(), **batch_np(toy), **sd_np(s32), **grads_np(s32))

    # ---- (iv) config 1: QM9-shaped bs 32 SchNet forward ---------------------------------
    torch.manual_seed(41)
    qb = make_qm9_batch(32, seed=3)
    schq = ns.SchNet(hidden_channels=32, num_filters=32, num_interactions=3, num_gaussians=51, cutoff=10,
                     readout="mean", node_class=119)
    outq, hq = schq(qb.x, qb.positions, qb.batch, return_latent=True)
    save("qm9_schnet.npz", out=outq.detach().numpy(), h=hq.detach().numpy(), **batch_np(qb), **sd_np(schq))

    # ---- (v) 20-step loss curve, bs 8, dropout off, seeded noise -------------------------
    torch.manual_seed(51)
    lb = make_batch(8, seed=5)
    gnn, sch, s23, s32 = build_ref(ns, 32, 32, 2, 51, layers=3)
    init = {}
    for nm, m in (("model_2D", gnn), ("model_3D", sch), ("SDE_2Dto3D_model", s23), ("SDE_3Dto2D_model", s32)):
        init.update(sd_np(m, nm + ".sd."))
    groups = [{"params": gnn.parameters(), "lr": 1e-3}, {"params": sch.parameters(), "lr": 1e-4},
              {"params": s23.parameters(), "lr": 1e-3}, {"params": s32.parameters(), "lr": 1e-4}]
    opt = torch.optim.Adam(groups, lr=1e-3, weight_decay=0.0)
    crit = torch.nn.BCEWithLogitsLoss()
    curve = []
    for step in range(20):
        torch.manual_seed(1000 + step)   # replayable noise source: program-order draws on the CPU generator
        bb = lb.clone()
        h2 = gnn(bb.x, bb.edge_index, bb.edge_attr)
        _, h3 = sch(bb.x[:, 0], bb.positions, bb.batch, return_latent=True)
        # dual_CL (util.py:52-68,76-79) restated inline: util.py imports rdkit at module top
        def cl(X, Y):
            neg = Y[torch.randperm(len(Y))]
            pp = torch.sum(X * Y, 1) / 0.1
            pn = torch.sum(X * neg, 1) / 0.1
            return crit(pp, torch.ones_like(pp)) + crit(pn, torch.zeros_like(pn))
        l_cl = (cl(h2, h3) + cl(h3, h2)) / 2
        l23 = s23(h2, bb, anneal_power=0)["position"]
        lx, la = s32(h3, bb, reduce_mean=True, continuous=True, train=True, anneal_power=0)
        l32 = (lx + la) * 0.5
        loss = l_cl + l23 + l32
        opt.zero_grad()
        loss.backward()
        opt.step()
        curve.append([loss.item(), l_cl.item(), l23.item(), l32.item()])
    save("losscurve.npz", curve=np.array(curve, dtype=np.float64), lrs=np.array([1e-3, 1e-4, 1e-3, 1e-4]),
         seed_base=np.int64(1000), **batch_np(lb), **init)

    # ---- sampler: 5 predictor-corrector steps, genuine VESDE.reverse().discretize ---------
    torch.manual_seed(61)
    gnn, sch, s23, s32 = build_ref(ns, TOY["emb"], TOY["filters"], TOY["interactions"], TOY["gaussians"])
    s23.eval(); gnn.eval()
    sb = make_batch(2, seed=13, sizes=[6, 6])
    with torch.no_grad():
        rep = gnn(sb.x, sb.edge_index, sb.edge_attr)
    sde = s23.sde_pos
    rsde = sde.reverse(s23, probability_flow=False)
    Nn = sb.x.size(0)
    pos = torch.randn(Nn, 3)
    pos0 = pos.clone()
    ts = torch.linspace(1.0, 1e-4, 1000)[:5]
    noises_p, noises_c, traj = [], [], []
    snr, seps, n_steps = 0.16, 0.7, 1
    for tval in ts:
        vec_t = torch.ones(Nn) * tval
        # corrector (LangevinCorrector.update_fn :191-212), VE: alpha = 1
        grad = s23.get_score(rep, sb, pos, None, vec_t)
        noise = torch.randn_like(pos)
        gn = torch.norm(grad.reshape(Nn, -1), dim=-1).mean()
        nn_ = torch.norm(noise.reshape(Nn, -1), dim=-1).mean()
        step = (snr * nn_ / gn) ** 2 * 2 * torch.ones_like(vec_t)
        pos = pos + step[:, None] * grad + torch.sqrt(step * 2)[:, None] * noise * seps
        noises_c.append(noise)
        # predictor (ReverseDiffusionPredictor.update_fn :163-168)
        f, G = rsde.discretize(pos, rep, sb, vec_t)
        noise = torch.randn_like(pos)
        pos = (pos - f) + G[:, None] * noise
        noises_p.append(noise)
        traj.append(pos.clone())
    save("sampler.npz", pos0=pos0.numpy(), ts=ts.numpy(), noise_pred=torch.stack(noises_p).numpy(),
         noise_corr=torch.stack(noises_c).numpy(), traj=torch.stack(traj).numpy(), snr=np.float64(snr),
         scale_eps=np.float64(seps), rep=rep.numpy(), **batch_np(sb), **sd_np(s23))

    # ---- (vi) inventory at the README configuration --------------------------------------
    torch.manual_seed(0)
    gnn = ns.GNN(5, 300, JK="last", drop_ratio=0, gnn_type="GIN")
    sch = ns.SchNet(hidden_channels=300, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=10,
                    readout="mean", node_class=119)
    s23 = ns.SDEModel2Dto3D_02(emb_dim=300, hidden_dim=32, beta_min=0.2, beta_max=1.0, num_diffusion_timesteps=1000,
                               beta_schedule=None, SDE_type="VE", use_extend_graph=True)
    s32 = ns.SDEModel3Dto2D_node_adj_dense(dim3D=300, c_init=2, c_hid=8, c_final=4, num_heads=4, adim=16, nhid=16,
                                           num_layers=4, emb_dim=300, num_linears=3, beta_min=0.1, beta_max=1.0,
                                           num_diffusion_timesteps=1000, SDE_type="VE", num_class_X=119,
                                           noise_on_one_hot=True)
    inv = {}
    for nm, m in (("model_2D", gnn), ("model_3D", sch), ("SDE_2Dto3D_model", s23), ("SDE_3Dto2D_model", s32)):
        inv[nm] = {
            "trainable": int(sum(p.numel() for p in m.parameters() if p.requires_grad)),
            "frozen": int(sum(p.numel() for p in m.parameters() if not p.requires_grad)),
            "state_dict": {k: [list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()},
        }
    with open(os.path.join(OUT, "inventory.json"), "w") as f:
        json.dump(inv, f, indent=0, sort_keys=False)
    print("inventory:", {k: v["trainable"] for k, v in inv.items()})


if __name__ == "__main__":
    assert ref_loader.available(), "needs /root/reference"
    main()
