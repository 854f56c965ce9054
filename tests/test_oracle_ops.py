"""Known-answer tests pinning the third-party op semantics the oracle restates.
These ops live in PyG / torch-scatter, not under /root/reference, so the reference cannot pin them:
hand-computed cases do."""
import math

import torch

from oracle import restate as R


def test_scatter_sum_mean_known_answer():
    src = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0], [7.0, 8.0]])
    idx = torch.tensor([2, 0, 2, 2])
    s = R.scatter_sum(src, idx, 4)
    assert s.tolist() == [[3.0, 4.0], [0.0, 0.0], [13.0, 16.0], [0.0, 0.0]]
    m = R.scatter_mean(src, idx, 4)
    # empty rows: sum 0 / clamp(count,1) = 0
    assert torch.allclose(m, torch.tensor([[3.0, 4.0], [0.0, 0.0], [13 / 3, 16 / 3], [0.0, 0.0]]))


def test_segment_softmax_rows_sum_to_one_and_known_answer():
    src = torch.tensor([[0.0], [math.log(3.0)], [5.0]])
    idx = torch.tensor([1, 1, 0])
    out = R.segment_softmax(src, idx, 3)
    assert torch.allclose(out.flatten(), torch.tensor([0.25, 0.75, 1.0]), atol=1e-6)
    big = torch.randn(50, 8) * 30
    idx = torch.randint(0, 7, (50,))
    sm = R.segment_softmax(big, idx, 7)
    sums = R.scatter_sum(sm, idx, 7)
    present = torch.bincount(idx, minlength=7) > 0
    assert torch.allclose(sums[present], torch.ones_like(sums[present]), atol=1e-5)


def test_radius_graph_known_answer():
    # two molecules; atom 3 is 11 A from atom 2 (outside r=10); strict inequality at exactly r
    pos = torch.tensor([[0.0, 0, 0], [1.0, 0, 0], [0.0, 0, 0], [11.0, 0, 0], [10.0, 0, 0]])
    batch = torch.tensor([0, 0, 1, 1, 1])
    ei = R.radius_graph(pos, 10.0, batch)
    edges = set(map(tuple, ei.t().tolist()))
    # (source, target); pair (2,4) is at exactly 10.0 -> excluded (strict <); (3,4) at 1.0 included
    assert edges == {(1, 0), (0, 1), (4, 3), (3, 4)}
    # grouped by target, ascending
    assert ei[1].tolist() == sorted(ei[1].tolist())


def test_radius_graph_neighbour_cap():
    pos = torch.zeros(40, 3)
    pos[:, 0] = torch.arange(40) * 0.01
    ei = R.radius_graph(pos, 10.0, torch.zeros(40, dtype=torch.long), max_num_neighbors=32)
    deg = torch.bincount(ei[1], minlength=40)
    assert int(deg.max()) == 32 and int(deg.min()) == 32


def test_to_dense_known_answer():
    batch = torch.tensor([0, 0, 1, 1, 1])
    x = torch.arange(5.0).view(5, 1) + 1
    d = R.to_dense_batch(x, batch, 3, 2)
    assert d[:, :, 0].tolist() == [[1.0, 2.0, 0.0], [3.0, 4.0, 5.0]]
    ei = torch.tensor([[0, 1, 2, 4, 4], [1, 0, 4, 2, 2]])
    ea = torch.tensor([1.0, 1.0, 2.0, 2.0, 3.0])
    adj = R.to_dense_adj(ei, batch, ea, 3, 2)
    assert adj[0].tolist() == [[0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 0.0]]
    assert adj[1].tolist() == [[0.0, 0.0, 2.0], [0.0, 0.0, 0.0], [5.0, 0.0, 0.0]]  # duplicates add


def test_gin_message_known_answer():
    # one GIN aggregation by hand: out_i = (1+eps) x_i + sum_j relu(x_j + e_ji)
    conv = R.GINConv(2)
    conv.mlp = torch.nn.Identity()
    with torch.no_grad():
        for emb in conv.bond_encoder.bond_embedding_list:
            emb.weight.zero_()
        conv.bond_encoder.bond_embedding_list[0].weight[1] = torch.tensor([-5.0, 1.0])
        conv.eps.fill_(0.5)
    x = torch.tensor([[1.0, 2.0], [3.0, -4.0]])
    ei = torch.tensor([[0, 1], [1, 0]])
    ea = torch.tensor([[1, 0, 0], [0, 0, 0]])
    out = conv(x, ei, ea)
    # target 1 gets relu(x0 + [-5,1]) = [0,3]; target 0 gets relu(x1 + 0) = [3,0]
    assert torch.allclose(out, torch.tensor([[1.5 + 3.0, 3.0 + 0.0], [4.5 + 0.0, -6.0 + 3.0]]))


def test_cfconv_known_answer():
    # 2 atoms 1 A apart, filters = 1: agg_i = x1_j * W_ij, W = mlp(rbf) * 0.5(cos(pi d / rc) + 1)
    conv = R.InteractionBlock(hidden_channels=1, num_gaussians=3, num_filters=1, cutoff=2.0)
    with torch.no_grad():
        conv.mlp[0].weight.fill_(1.0); conv.mlp[0].bias.zero_()
        conv.mlp[2].weight.fill_(2.0); conv.mlp[2].bias.fill_(0.5)
        conv.conv.lin1.weight.fill_(3.0)
        conv.conv.lin2.weight.fill_(1.0); conv.conv.lin2.bias.zero_()
    x = torch.tensor([[1.0], [2.0]])
    ei = torch.tensor([[0, 1], [1, 0]])
    d = torch.tensor([1.0, 1.0])
    rbf = R.GaussianSmearing(0.0, 2.0, 3)(d)
    expect_rbf = torch.exp(-0.5 * (d.view(-1, 1) - torch.tensor([0.0, 1.0, 2.0])) ** 2)
    assert torch.allclose(rbf, expect_rbf)
    s = float(expect_rbf[0].sum())
    ssp = math.log1p(math.exp(s)) - math.log(2.0)
    W = (2.0 * ssp + 0.5) * 0.5 * (math.cos(math.pi * 1.0 / 2.0) + 1.0)
    out = conv.conv(x, ei, d, rbf)
    assert torch.allclose(out.flatten(), torch.tensor([3.0 * 2.0 * W, 3.0 * 1.0 * W]), atol=1e-6)


def test_transformer_conv_single_edge_known_answer():
    # one incoming edge -> softmax weight 1 -> out_i = (W_v x_j + W_e e) + W_skip x_i
    tc = R.TransformerConv(4, 2, 2, 0.0, 4)
    x = torch.randn(2, 4)
    e = torch.randn(1, 4)
    out = tc(x, torch.tensor([[0], [1]]), e)
    exp1 = tc.lin_value(x[0]) + tc.lin_edge(e[0]) + tc.lin_skip(x[1])
    exp0 = tc.lin_skip(x[0])
    assert torch.allclose(out[1], exp1, atol=1e-6) and torch.allclose(out[0], exp0, atol=1e-6)


def test_vesde_known_answers():
    ve = R.VESDE(0.2, 1.0, 1000)
    _, std = ve.marGINal_prob(torch.zeros(3, 3), torch.tensor([1e-6, 0.5, 1.0]))
    assert torch.allclose(std, torch.tensor([0.2, 0.2 * 5 ** 0.5, 1.0]), atol=1e-5)
    _, G = ve.discretize(torch.zeros(1, 3), torch.tensor([0.5]))
    assert abs(float(G) - 0.025345) < 1e-5        # SURVEY.md App. C.5 known answer


def test_transformer_conv_multi_head_multi_edge_hand_computed():
    """PyG 2.0.2 TransformerConv semantics (App. A.4) on a case small enough to evaluate by hand with explicit loops:
    2 heads x 2 channels, node 2 receives THREE edges (two from the same source: duplicate edges are separate
    messages), node 3 receives none (in-degree 0 -> only lin_skip), the edge term enters BOTH key and value, scores are
    scaled by 1/sqrt(C) and normalised per (target, head) with the +1e-16 denominator."""
    torch.manual_seed(0)
    H, C, Fin, Fe = 2, 2, 3, 2
    tc = R.TransformerConv(Fin, C, H, 0.0, Fe)
    x = torch.randn(4, Fin)
    ei = torch.tensor([[0, 1, 0, 2], [2, 2, 2, 1]])       # edges 0->2, 1->2, 0->2 (duplicate), 2->1
    ea = torch.randn(4, Fe)
    out = tc(x, ei, ea)
    q, k, v, s = tc.lin_query(x), tc.lin_key(x), tc.lin_value(x), tc.lin_skip(x)
    e = tc.lin_edge(ea)
    exp = s.clone()
    for tgt in range(4):
        inc = [m for m in range(ei.size(1)) if int(ei[1, m]) == tgt]
        for h in range(H):
            sl = slice(h * C, (h + 1) * C)
            scores = [float((q[tgt, sl] * (k[int(ei[0, m]), sl] + e[m, sl])).sum()) / math.sqrt(C) for m in inc]
            if not inc:
                continue
            mx = max(scores)
            w = [math.exp(sc - mx) for sc in scores]
            den = sum(w) + 1e-16
            for m, wm in zip(inc, w):
                exp[tgt, sl] = exp[tgt, sl] + (v[int(ei[0, m]), sl] + e[m, sl]) * (wm / den)
    assert torch.allclose(out, exp, atol=1e-6), (out - exp).abs().max()
    assert torch.allclose(out[3], s[3]) and torch.allclose(out[0], s[0])          # in-degree 0: skip connection only
    # the two duplicate edges 0->2 are two messages: removing one changes the result
    out2 = tc(x, ei[:, [0, 1, 3]], ea[[0, 1, 3]])
    assert not torch.allclose(out2[2], out[2])


def test_to_dense_adj_duplicates_and_cross_molecule_offsets():
    """PyG to_dense_adj (App. A.6): duplicate edges ADD, local indices are relative to each molecule's first atom, and
    entries of the padded rows / columns stay zero."""
    batch = torch.tensor([0, 0, 0, 1, 1])
    ei = torch.tensor([[0, 0, 0, 2, 3, 4, 4], [1, 1, 1, 0, 4, 3, 3]])
    ea = torch.tensor([1.0, 2.0, 4.0, 8.0, 16.0, 32.0, 64.0])
    adj = R.to_dense_adj(ei, batch, ea, 3, 2)
    assert adj[0].tolist() == [[0.0, 7.0, 0.0], [0.0, 0.0, 0.0], [8.0, 0.0, 0.0]]
    assert adj[1].tolist() == [[0.0, 16.0, 0.0], [96.0, 0.0, 0.0], [0.0, 0.0, 0.0]]
