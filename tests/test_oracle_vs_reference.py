"""Live check (build container only): oracle/restate.py vs the reference's files executed verbatim.
Skipped wherever /root/reference is absent (the GPU box)."""
import warnings

import pytest
import torch

from oracle import ref_loader, restate as R
from moleculesde_amd.synthetic import make_batch

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not present")


def _pairs(E, filters, inter):
    ns = ref_loader.verbatim()
    kw23 = dict(emb_dim=E, hidden_dim=32, beta_min=0.2, beta_max=1.0, num_diffusion_timesteps=1000, beta_schedule=None,
                SDE_type="VE", use_extend_graph=True)
    kw32 = dict(dim3D=E, c_init=2, c_hid=8, c_final=4, num_heads=4, adim=16, nhid=16, num_layers=4, emb_dim=E,
                num_linears=3, beta_min=0.1, beta_max=1.0, num_diffusion_timesteps=1000, SDE_type="VE", num_class_X=119,
                noise_on_one_hot=True)
    kws = dict(hidden_channels=E, num_filters=filters, num_interactions=inter, num_gaussians=51, cutoff=10,
               readout="mean", node_class=119)
    ref = dict(gnn=ns.GNN(5, E, JK="last", drop_ratio=0, gnn_type="GIN"), sch=ns.SchNet(**kws),
               s23=ns.SDEModel2Dto3D_02(**kw23), s32=ns.SDEModel3Dto2D_node_adj_dense(**kw32))
    mine = dict(gnn=R.GNN(5, E, JK="last", drop_ratio=0, gnn_type="GIN"), sch=R.SchNet(**kws),
                s23=R.SDEModel2Dto3D_02(**kw23), s32=R.SDEModel3Dto2D_node_adj_dense(**kw32))
    for k in ref:
        sd = ref[k].state_dict()
        assert list(sd.keys()) == list(mine[k].state_dict().keys())
        mine[k].load_state_dict(sd)
    return ref, mine


def _step(m, b, seed):
    torch.manual_seed(seed)
    h2 = m["gnn"](b.x, b.edge_index, b.edge_attr)
    _, h3 = m["sch"](b.x[:, 0], b.positions, b.batch, return_latent=True)
    l23 = m["s23"](h2, b, anneal_power=0)["position"]
    lx, la = m["s32"](h3, b, reduce_mean=True, continuous=True, train=True, anneal_power=0)
    (l23 + lx + la).backward()
    return h2, h3, l23, lx, la


def test_full_step_matches_reference_with_dropout_on():
    """Same seed => same dropout masks and noise (program-order draws): forward is bit-identical."""
    warnings.filterwarnings("ignore")
    torch.manual_seed(0)
    ref, mine = _pairs(48, 32, 3)
    b = make_batch(8, 1)
    r = _step(ref, b.clone(), 5)
    o = _step(mine, b.clone(), 5)
    for a, c in zip(r, o):
        assert torch.allclose(a, c, rtol=1e-6, atol=1e-6)
    for k in ref:
        scale = max(p.grad.abs().max().item() for p in ref[k].parameters() if p.grad is not None)
        for (n, p), (_, q) in zip(ref[k].named_parameters(), mine[k].named_parameters()):
            if p.grad is None:
                assert q.grad is None
                continue
            assert (p.grad - q.grad).abs().max().item() <= 1e-5 * scale, (k, n)


def test_standin_semantics_equal_oracle_ops():
    """The stand-in layer used to run the reference and the oracle's own op restatements agree."""
    ns = ref_loader.verbatim()
    torch.manual_seed(1)
    b = make_batch(6, 2)
    ei_s = ns.standins.nn.radius_graph(b.positions, r=3.0, batch=b.batch)
    ei_o = R.radius_graph(b.positions, 3.0, b.batch)
    assert torch.equal(ei_s, ei_o)
    src = torch.randn(ei_o.size(1), 4)
    assert torch.allclose(ns.standins.utils.softmax(src, ei_o[1], None, b.x.size(0)),
                          R.segment_softmax(src, ei_o[1], b.x.size(0)))
    assert torch.allclose(ns.standins.scatter.scatter(src, ei_o[1], dim=0, dim_size=b.x.size(0), reduce="mean"),
                          R.scatter_mean(src, ei_o[1], b.x.size(0)))
