"""Kernel orchestration of the 3D->2D dense score head on RAGGED data.

`dense_head_losses` is the product path of `SDEModel3Dto2D_node_adj_dense.forward`
(SDE_model_3D_to_2D_node_adj_dense.py:101-179): ONE autograd node whose forward and backward launch only kernels of
libmsde_hip.so -- csrc/dense_head.hip (per-molecule kernels) and csrc/gemm_ex.hip / the grouped weight-gradient kernel
(everything GEMM shaped).  No padding, no torch operator inside.

Layout (see csrc/dense_head.hip): atoms in batch order; pairs of molecule b at rows pair_ptr[b] + i*n_b + j.
  XC [N, F+64]   columns 0..F-1: x = embedding_3D(h) + embedding_X(perturbed one-hot) (:156); F..F+63: the four dense-GCN
                 outputs of the node network -- i.e. exactly cat(x_list) of invariant_scorenetwork_dense.py:123-124.
                 SDEModel3Dto2D_node_adj_dense_02 (:326) concatenates the two embeddings instead: x is 2 F wide (XC
                 [N, 2F+64]), the two embedding products write adjacent column blocks
  AC [P, 32]     columns 0-1 perturbed adjacency and its square, 2..29 the edge layers' outputs = the concatenation the
                 final pair MLP reads (:81-84)
"""
import ctypes
import types

import torch

from .. import _lib, hip, slabs, wcache

# the C per-channel weight gradients of an edge layer's node branch as ONE product x^T gXV (False: C products with 16-wide operands);
# --full 3.209 vs 3.211 ms (profiles/r06_ab_merge_channel_wgrad.txt): 16 problems fewer in the grouped launch, the same time
MERGE_CHANNEL_WGRAD = True

AC_LD, XP_LD = 32, 120
_p = hip._p


def _empty(*shape, device):
    return torch.empty(*shape, dtype=torch.float32, device=device)


def ragged_layout(mol_ptr_i32, n_max=None):
    """pair_ptr (int32 [B+1], device) and sizes for a batch whose atoms are grouped per molecule."""
    cnt = (mol_ptr_i32[1:] - mol_ptr_i32[:-1]).to(torch.int64)
    pair_ptr = torch.cat([cnt.new_zeros(1), (cnt * cnt).cumsum(0)]).to(torch.int32)
    return pair_ptr


# ------------------------------------------------------------------------------------------------ parameter packing
EDGE_KEYS = ("Wqk0", "bqk0", "Wqk1", "bqk1", "Wv", "bv", "m0W", "m0b", "m1W", "m1b", "m2W", "m2b", "c0W", "c0b", "c1W",
             "c1b")


def edge_layer_tensors(layer):
    """The 16 operands of one EdgeNetwork_dense layer; stacked ones are free views once FlatAdam laid them out."""
    at = list(layer.attn)
    cat = hip.cat_params
    return [cat([a.func_q.layers[0].weight for a in at] + [a.func_k.layers[0].weight for a in at]),
            cat([a.func_q.layers[0].bias for a in at] + [a.func_k.layers[0].bias for a in at]),
            cat([a.func_q.layers[1].weight for a in at] + [a.func_k.layers[1].weight for a in at]),
            cat([a.func_q.layers[1].bias for a in at] + [a.func_k.layers[1].bias for a in at]),
            cat([a.func_v.weight for a in at]), cat([a.func_v.bias for a in at]),
            layer.mlp.layers[0].weight, layer.mlp.layers[0].bias, layer.mlp.layers[1].weight, layer.mlp.layers[1].bias,
            layer.mlp.layers[2].weight, layer.mlp.layers[2].bias,
            layer.multi_channel.layers[0].weight, layer.multi_channel.layers[0].bias,
            layer.multi_channel.layers[1].weight, layer.multi_channel.layers[1].bias]


def edge_net_tensors(net):
    out = []
    for layer in net.layers:
        out += edge_layer_tensors(layer)
    for lin in net.final.layers:
        out += [lin.weight, lin.bias]
    return out


def node_net_tensors(net):
    cat = hip.cat_params
    out = [net.layers[0].weight, cat([l.weight for l in net.layers[1:]]), cat([l.bias for l in net.layers])]
    for lin in net.final.layers:
        out += [lin.weight, lin.bias]
    return out


def edge_net_shape(net):
    """[(C_in, C_out)] per layer and the channel offsets in AC."""
    chans = [(len(layer.attn), layer.mlp.layers[-1].weight.size(0)) for layer in net.layers]
    offs, o = [], 0
    for C, CO in chans:
        offs.append((o, o + C))
        o += C
    return chans, offs


def fused_supported(edge, node, n_max):
    """The kernels cover the configuration every MoleculeSDE script uses (pretrain_MoleculeSDE.py:310-315):
    nhid = adim = 16, num_linears = 3, c_init = 2, channel counts (2|8 -> 8|4), 4 GCN layers, <= 32 atoms."""
    try:
        chans, offs = edge_net_shape(edge)
        ok = all((C, CO) in ((2, 8), (8, 8), (8, 4)) for C, CO in chans) and offs[-1][1] + chans[-1][1] <= AC_LD - 2
        for i, layer in enumerate(edge.layers):
            a = layer.attn[0]
            ok = ok and a.func_q.layers[0].weight.size(0) == 32 and len(a.func_q.layers) == 2 and a.out_dim == 16
            ok = ok and len(layer.mlp.layers) == 3 and layer.mlp.layers[0].weight.size(0) == 16
            ok = ok and len(layer.multi_channel.layers) == 2 and layer.multi_channel.layers[0].weight.size(0) == 16
            if i > 0:
                ok = ok and chans[i][0] == chans[i - 1][1]
        ok = ok and edge.c_init == 2 and len(edge.final.layers) == 3 and edge.final.layers[2].weight.size(0) == 1
        ok = ok and node.depth == 4 and node.nhid == 16 and len(node.final.layers) == 3 and node.nout <= XP_LD
        return bool(ok) and n_max <= 32
    except Exception:
        return False


def _edge_struct(t):
    s = _lib.EdgeLayerParams()
    s.bv = t[5].data_ptr()
    s.mW0, s.mb0, s.mW1, s.mb1, s.mW2, s.mb2 = (x.data_ptr() for x in t[6:12])
    s.cW0, s.cb0, s.cW1, s.cb1 = (x.data_ptr() for x in t[12:16])
    return s


# ------------------------------------------------------------------------------------------------ edge network
def edge_forward(cfg, x0, AC, flags, chans, offs, T):
    """x0 [N, F] (may be a column block), AC columns 0..1 filled.  T: edge_net_tensors.  Returns the saved state; the
    pair MLP's last hidden layer (G2, ZG2) is in it (its Linear(60, 1) belongs to the loss kernel)."""
    dev, N, P, B = x0.device, cfg.N, cfg.P, cfg.B
    sv = types.SimpleNamespace(layers=[], x0=x0)
    x = x0
    for l, (C, CO) in enumerate(chans):
        t = T[16 * l:16 * l + 16]
        F = x.size(1)
        W2 = 2 * C * 32
        H, QK, XV = _empty(N, W2, device=dev), _empty(N, W2, device=dev), _empty(N, 16 * C, device=dev)
        hip.gemm_ex(x, t[0], H, bias=t[1], act="tanh")
        hip.gemm_ex(H, t[2], QK, bias=t[3], groups=2 * C, group_strides=dict(a=32, b=1024, bias=32, c=32), N=32, K=32)
        hip.gemm_ex(x, t[4], XV, b_kmajor=True, groups=C, group_strides=dict(b=F * 16, c=16), N=16, K=F)
        xo, IN = _empty(N, 16, device=dev), _empty(P, 2 * C, device=dev)
        H1, H2 = _empty(P, 16, device=dev), _empty(P, 16, device=dev)
        xcat, Hmc = _empty(N, 16 * C, device=dev), _empty(N, 16, device=dev)
        st = _edge_struct(t)
        _lib.call("msde_dense_edge_layer_fwd", _p(QK), _p(XV), _p(AC), offs[l][0], offs[l][1], C, CO, _p(flags),
                  _p(cfg.mol_ptr), _p(cfg.pair_ptr), ctypes.byref(st), B, cfg.n_max, _p(xo), _p(IN), _p(H1), _p(H2), _p(xcat),
                  _p(Hmc), hip._stream())
        sv.layers.append(types.SimpleNamespace(x=x, H=H, QK=QK, XV=XV, xo=xo, IN=IN, H1=H1, H2=H2, xcat=xcat, Hmc=Hmc))
        x = xo
    f = T[16 * len(chans):]
    fdim = f[0].size(1)
    sv.fdim = fdim
    w1, w2 = f[0].size(0), f[2].size(0)
    sv.ZG1, sv.G1, sv.ZG2, sv.G2 = (_empty(P, w, device=dev) for w in (w1, w1, w2, w2))
    hip.gemm_ex(AC[:, :fdim], f[0], sv.G1, bias=f[1], act="silu", Z=sv.ZG1)
    hip.gemm_ex(sv.G1, f[2], sv.G2, bias=f[3], act="silu", Z=sv.ZG2)
    return sv


def edge_backward(cfg, sv, AC, flags, chans, offs, T, gS, gZG2, gx0, gx0_accumulate, need_gadj0=False, gAC_out=None):
    """gS [P] / gZG2 [P, 60]: gradients of the pair MLP's scalar output / last hidden pre-activation.  Adds (or writes)
    the gradient of x0 into gx0.  Returns the parameter gradients in edge_net_tensors order."""
    dev, N, P, B = gZG2.device, cfg.N, cfg.P, cfg.B
    L = len(chans)
    f = T[16 * L:]
    G = [None] * len(T)
    wg = slabs.weight_grad
    G[16 * L + 4], G[16 * L + 5] = wg(gS.view(P, 1), sv.G2, True)
    G[16 * L + 2], G[16 * L + 3] = wg(gZG2, sv.G1, True)
    gZG1 = _empty(P, sv.ZG1.size(1), device=dev)
    hip.gemm_ex(gZG2, f[2], gZG1, b_kmajor=True, act="silu", dact_from=sv.ZG1)
    G[16 * L], G[16 * L + 1] = wg(gZG1, AC[:, :sv.fdim], True)
    gAC = _empty(P, AC_LD, device=dev) if gAC_out is None else gAC_out
    hip.gemm_ex(gZG1, f[0], gAC[:, :sv.fdim], b_kmajor=True)
    g_xnext = None
    for l in range(L - 1, -1, -1):
        C, CO = chans[l]
        t = T[16 * l:16 * l + 16]
        s = sv.layers[l]
        F = s.x.size(1)
        W2 = 2 * C * 32
        node = g_xnext is not None
        gQK = _empty(N, W2, device=dev)
        GO, GH2, GH1 = _empty(P, CO, device=dev), _empty(P, 16, device=dev), _empty(P, 16, device=dev)
        gXV = GY = GHm = GV = None
        if node:
            gXV, GV = _empty(N, 16 * C, device=dev), _empty(N, 16 * C, device=dev)
            GY, GHm = _empty(N, 16, device=dev), _empty(N, 16, device=dev)
        st = _edge_struct(t)
        need_gadj = 1 if (l > 0 or need_gadj0) else 0
        _lib.call("msde_dense_edge_layer_bwd", _p(s.QK), _p(s.XV), _p(AC), _p(gAC), offs[l][0], offs[l][1], C, CO, _p(flags),
                  _p(cfg.mol_ptr), _p(cfg.pair_ptr), ctypes.byref(st), B, cfg.n_max, _p(s.xo), _p(g_xnext), _p(s.IN), _p(s.H1),
                  _p(s.H2), _p(s.xcat), _p(s.Hmc), need_gadj, _p(gQK), _p(gXV), _p(GO), _p(GH2), _p(GH1), _p(GY), _p(GHm),
                  _p(GV), hip._stream())
        b = 16 * l
        G[b + 10], G[b + 11] = wg(GO, s.H2, True)
        G[b + 8], G[b + 9] = wg(GH2, s.H1, True)
        G[b + 6], G[b + 7] = wg(GH1, s.IN, True)
        if node:
            G[b + 14], G[b + 15] = wg(GY, s.Hmc, True)
            G[b + 12], G[b + 13] = wg(GHm, s.xcat, True)
            G[b + 5] = slabs.colsum_leaf(GV)
            gWv = _empty(C * F, 16, device=dev)
            # x W_c with W_c stored [in, out]: gW_c = x^T g(xW_c) -- the C channels share x, and their g(xW_c) are adjacent column
            # blocks of gXV: ONE product x^T gXV [F, 16 C] whose column block c is channel c's gradient (the slab reduction
            # writes the blocks in place) instead of C products with 16-wide operands (half-used matrix-core tiles)
            if MERGE_CHANNEL_WGRAD:
                slabs.weight_grad_blocks(s.x, gXV, False, [(16 * c, 16, gWv[c * F:(c + 1) * F], 0) for c in range(C)])
            else:
                for c in range(C):
                    wg(s.x, gXV[:, 16 * c:16 * c + 16], False, out_w=gWv[c * F:(c + 1) * F])
            G[b + 4] = gWv
        # second q/k layers (block diagonal) and the gradient of their tanh hidden layer
        gH = _empty(N, W2, device=dev)
        hip.gemm_ex(gQK, t[2], gH, b_kmajor=True, groups=2 * C, group_strides=dict(a=32, b=1024, c=32, r=32), N=32, K=32,
                    act="tanh", dact_from=s.H)
        gW1, gb1 = _empty(W2, 32, device=dev), _empty(W2, device=dev)
        for g in range(2 * C):
            wg(gQK[:, 32 * g:32 * g + 32], s.H[:, 32 * g:32 * g + 32], True, out_w=gW1[32 * g:32 * g + 32],
               out_b=gb1[32 * g:32 * g + 32])
        G[b + 2], G[b + 3] = gW1, gb1
        G[b], G[b + 1] = wg(gH, s.x, True)
        # gradient of the layer input x
        if l > 0:
            gx = _empty(N, 16, device=dev)
            hip.gemm_ex(gH, t[0], gx, b_kmajor=True)
            dst, acc = gx, True
        else:
            hip.gemm_ex(gH, t[0], gx0, b_kmajor=True, accumulate=gx0_accumulate)
            dst, acc = gx0, True
        if node:
            hip.gemm_ex(gXV, t[4], dst, accumulate=acc, N=F, K=16 * C, b_kblk=(16, F * 16, 16))
        g_xnext = dst if l > 0 else None
    return G, gAC


# ------------------------------------------------------------------------------------------------ node network
def node_forward(cfg, XC, F, AC, T):
    """XC [N, F+64] with x in columns 0..F-1; fills columns F.. with the GCN outputs and runs the final MLP."""
    dev, N, B = XC.device, cfg.N, cfg.B
    sv = types.SimpleNamespace()
    X = XC[:, :F]
    sv.XW0 = _empty(N, 16, device=dev)
    hip.gemm_ex(X, T[0], sv.XW0, b_kmajor=True, N=16, K=F)
    ld = XC.stride(0)
    _lib.call("msde_dense_node_gcn_fwd", _p(sv.XW0), _p(AC), _p(cfg.mol_ptr), _p(cfg.pair_ptr), _p(T[1]), _p(T[2]), B, cfg.n_max,
              ctypes.c_void_p(XC.data_ptr() + 4 * F), ld, hip._stream())
    w1, w2, nout = T[3].size(0), T[5].size(0), T[7].size(0)
    sv.Z1, sv.F1, sv.Z2, sv.F2 = (_empty(N, w, device=dev) for w in (w1, w1, w2, w2))
    sv.OUT = _empty(N, XP_LD, device=dev)
    node_mlp_forward(XC, T[3], T[4], T[5], T[6], T[7], T[8], sv.Z1, sv.F1, sv.Z2, sv.F2, sv.OUT)
    sv.nout = nout
    return sv


def _rs_ok(M, *tensors):
    return 0 < M <= hip.RS_MAX_ROWS and all(t.is_leaf and t.is_contiguous() and t.dtype == torch.float32 for t in tensors)


def node_mlp_forward(XC, W1, b1, W2, b2, W3, b3, Z1, F1, Z2, F2, OUT):
    """NodeScoreNetwork_dense.final (invariant_scorenetwork_dense.py:126-127): 364 -> 728 -> 728 -> 119 with SiLU, bias + SiLU +
    pre-activation store in the epilogues.  Layer 1 on msde_gemm_ex (64 x 64 tiles win at K = 364: 29.7 vs 33.3 us), layer 2
    on the row-strip kernel (46 vs 51 us), the 119-wide output layer on the row-strip kernel against a cached transposed copy
    of its weight padded to 120 columns (OUT has that row stride anyway): 12 us instead of 36 us for a 64-wide tile grid that
    is 7 % full in its last column block."""
    M, nout, w2 = XC.size(0), W3.size(0), W2.size(0)
    hip.gemm_ex(XC, W1, F1, bias=b1, act="silu", Z=Z1)
    hip.gemm_fwd(F1, W2, F2, bias=b2, act="silu", Z=Z2)
    if _rs_ok(M, W3, b3) and w2 % 4 == 0 and nout <= XP_LD:
        Wt = wcache.weight_layout("dense_node_out_t", (W3,), (w2, XP_LD), [(W3, 0, 0, nout, w2, 0, 0, True)])
        bp = wcache.weight_layout("dense_node_out_b", (b3,), (XP_LD,), [(b3, 0, 0, 1, nout, 0, 0, False)])
        hip.gemm_rs(F2, Wt, OUT, bias=bp, b_kmajor=True, N=XP_LD, K=w2, fallback=False)
    else:
        hip.gemm_ex(F2, W3, OUT[:, :nout], bias=b3)


def node_backward(cfg, sv, XC, F, AC, T, gOUT):
    """gOUT [N, 120] (columns < nout used).  Returns (parameter gradients in node_net_tensors order, gXC [N, F+64] whose
    first F columns are the gradient of x)."""
    dev, N, B = XC.device, cfg.N, cfg.B
    G = [None] * len(T)
    wg = slabs.weight_grad
    nout = sv.nout
    go = gOUT[:, :nout]
    G[7], G[8] = wg(go, sv.F2, True)
    gZ2 = _empty(N, sv.Z2.size(1), device=dev)
    if _rs_ok(N, T[7]) and sv.Z2.size(1) % 4 == 0 and gOUT.size(1) == XP_LD and nout <= XP_LD:
        # K = 119 -> 120: the gradient's padding column is zero (dense_loss_bwd writes it), the weight's padding row too
        Wp = wcache.weight_layout("dense_node_out_pad", (T[7],), (XP_LD, T[7].size(1)), [(T[7], 0, 0, nout, T[7].size(1), 0, 0, False)])
        hip.gemm_rs(gOUT, Wp, gZ2, b_kmajor=True, N=T[7].size(1), K=XP_LD, act="silu", dact_from=sv.Z2, fallback=False)
    else:
        hip.gemm_ex(go, T[7], gZ2, b_kmajor=True, act="silu", dact_from=sv.Z2)
    G[5], G[6] = wg(gZ2, sv.F1, True)
    gZ1 = _empty(N, sv.Z1.size(1), device=dev)
    hip.gemm_dgrad(gZ2, T[5], gZ1, act="silu", dact_from=sv.Z1)
    G[3], G[4] = wg(gZ1, XC, True)
    gXC = _empty(N, XC.size(1), device=dev)
    hip.gemm_ex(gZ1, T[3], gXC, b_kmajor=True)
    GP, MM = _empty(N, 64, device=dev), _empty(N, 64, device=dev)
    ld = XC.stride(0)
    _lib.call("msde_dense_node_gcn_bwd", ctypes.c_void_p(gXC.data_ptr() + 4 * F), gXC.stride(0),
              ctypes.c_void_p(XC.data_ptr() + 4 * F), ld, _p(AC), _p(cfg.mol_ptr), _p(cfg.pair_ptr), _p(T[1]), B, cfg.n_max,
              _p(GP), _p(MM), hip._stream())
    G[2] = slabs.colsum_leaf(GP)
    gWl = _empty(48, 16, device=dev)
    for l in range(1, 4):       # W_l stored [in, out]: gW_l = x_l^T (An^T g_pre_l)
        wg(XC[:, F + 16 * (l - 1):F + 16 * l], MM[:, 16 * l:16 * l + 16], False, out_w=gWl[16 * (l - 1):16 * l])
    G[1] = gWl
    G[0], _ = wg(XC[:, :F], MM[:, :16], False)
    hip.gemm_ex(MM[:, :16], T[0], gXC[:, :F], accumulate=True, N=F, K=16)       # g x += g(xW_0) W_0^T
    return G, gXC


# ------------------------------------------------------------------------------------------------ the product op
class _DenseHeadLosses(torch.autograd.Function):
    """(loss_x, loss_adj) of SDEModel3Dto2D_node_adj_dense.forward as one autograd node: h3 and the 4 + E + M parameter
    operands in, a 2-vector out."""

    @staticmethod
    def forward(ctx, cfg, h3, *T):
        dev = h3.device
        N, P, B, F = cfg.N, cfg.P, cfg.B, h3.size(1)
        nE = cfg.n_edge_tensors
        W3, b3, WX, bX = T[:4]
        TE, TN = T[4:4 + nE], T[4 + nE:]
        h3 = hip._f32(h3)
        AC, z_adj = _empty(P, AC_LD, device=dev), _empty(P, device=dev)
        flags, mean_std = _empty(N, device=dev), _empty(B, 2, device=dev)
        px, z_x = _empty(N, XP_LD, device=dev), _empty(N, XP_LD, device=dev)
        _lib.call("msde_dense_prepare", _p(cfg.bond_rowptr), _p(cfg.bond_src), _p(cfg.bond_val), _p(cfg.z_atom),
                  _p(cfg.mol_ptr), _p(cfg.pair_ptr), _p(cfg.draws), _p(cfg.t_in), B, cfg.T, cfg.eps, cfg.sde_vp, cfg.p0,
                  cfg.p1, _p(cfg.noise_adj), _p(cfg.noise_x), cfg.nm_pad, cfg.seed, _p(cfg.seed_dev), cfg.ncls, cfg.n_max,
                  _p(AC), _p(z_adj), _p(flags), _p(mean_std), _p(px), _p(z_x), hip._stream())
        # x = embedding_3D(h) + embedding_X(x) (:156), or for SDEModel3Dto2D_node_adj_dense_02 their CONCATENATION (:326): the
        # second product then writes the next F columns of XC instead of accumulating into the first -- the score networks
        # read a 2 F-wide x, nothing else changes.  FX = width of x.
        concat = bool(getattr(cfg, "concat", False))
        FX = 2 * F if concat else F
        XC = _empty(N, FX + 64, device=dev)
        # Two launches, not one two-segment product: embedding_X's weight rows are 119 floats (not 16-byte aligned) and would
        # put the big product on the scalar-load path too
        if concat:
            hip.gemm_ex(h3, W3, XC[:, :F], bias=b3)
            hip.gemm_ex(px[:, :cfg.ncls], WX, XC[:, F:FX], bias=bX)
        else:
            hip.gemm_ex(h3, W3, XC[:, :F], bias=b3, bias2=bX)
            hip.gemm_ex(px[:, :cfg.ncls], WX, XC[:, :F], accumulate=True)
        chans, offs = cfg.chans, cfg.offs
        se = edge_forward(cfg, XC[:, :FX], AC, flags, chans, offs, TE)
        sn = node_forward(cfg, XC, FX, AC, TN)
        f2W, f2b = TE[-2], TE[-1]
        res_adj, res_x = _empty(P, device=dev), _empty(N, XP_LD, device=dev)
        part, out = _empty(B * 4, 2, device=dev), _empty(2, device=dev)      # MSDE_DENSE_LOSS_SPLITS partials per molecule
        _lib.call("msde_dense_loss_fwd", _p(se.G2), se.G2.size(1), _p(f2W), _p(f2b), _p(sn.OUT), _p(z_adj), _p(z_x), _p(flags),
                  _p(mean_std), _p(cfg.mol_ptr), _p(cfg.pair_ptr), B, cfg.ncls, cfg.anneal, cfg.scale_x, cfg.scale_adj,
                  _p(getattr(cfg, "nmax_dev", None)), _p(res_adj), _p(res_x), _p(part), _p(out), hip._stream())
        ctx.cfg, ctx.se, ctx.sn = cfg, se, sn
        ctx.keep = (h3, AC, flags, mean_std, px, XC, res_adj, res_x)
        ctx.T = T
        # two 0-d outputs (views of one buffer): the caller's `loss_x, loss_adj = ...` then costs no operator, and the
        # backward receives the two upstream scalars separately
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_lx, g_la):
        cfg, se, sn, T = ctx.cfg, ctx.se, ctx.sn, ctx.T
        h3, AC, flags, mean_std, px, XC, res_adj, res_x = ctx.keep
        dev = h3.device
        N, P, B, F = cfg.N, cfg.P, cfg.B, h3.size(1)
        nE = cfg.n_edge_tensors
        W3, b3, WX, bX = T[:4]
        TE, TN = T[4:4 + nE], T[4 + nE:]
        g_lx = hip._f32(g_lx) if g_lx is not None else None
        g_la = hip._f32(g_la) if g_la is not None else None
        gS, gZG2 = _empty(P, device=dev), _empty(P, se.ZG2.size(1), device=dev)
        gOUT = _empty(N, XP_LD, device=dev)
        _lib.call("msde_dense_loss_bwd", _p(g_lx), _p(g_la), _p(res_adj), _p(res_x), _p(se.ZG2), se.ZG2.size(1), _p(TE[-2]), _p(flags),
                  _p(mean_std), _p(cfg.mol_ptr), _p(cfg.pair_ptr), B, cfg.ncls, cfg.anneal, cfg.scale_x, cfg.scale_adj,
                  _p(getattr(cfg, "nmax_dev", None)), _p(gS), _p(gZG2), _p(gOUT), hip._stream())
        concat = bool(getattr(cfg, "concat", False))
        FX = 2 * F if concat else F
        GN, gXC = node_backward(cfg, sn, XC, FX, AC, TN, gOUT)
        GE, _ = edge_backward(cfg, se, AC, flags, cfg.chans, cfg.offs, TE, gS, gZG2, gXC[:, :FX], True)
        gX = gXC[:, :F]                                   # gradient of embedding_3D's output
        gXx = gXC[:, F:FX] if concat else gX              # ... of embedding_X's (the same tensor when they were added)
        gW3, gb3 = slabs.weight_grad(gX, h3, True)
        gWX, gbX = slabs.weight_grad(gXx, px[:, :cfg.ncls], True)
        g_h3 = None
        if ctx.needs_input_grad[1]:
            g_h3 = _empty(N, F, device=dev)
            hip.gemm_ex(gX, W3, g_h3, b_kmajor=True)
        ctx.se = ctx.sn = ctx.keep = None
        return (None, g_h3, gW3, gb3, gWX, gbX) + tuple(GE) + tuple(GN)


FUSED_CALLS = 0     # incremented per fused forward: lets tests assert that the kernel path (not the operator path) ran


def dense_head_losses(cfg, h3, T):
    global FUSED_CALLS
    FUSED_CALLS += 1
    return _DenseHeadLosses.apply(cfg, h3, *T)


# ------------------------------------------------------------------------------------------------ stand-alone nets
class _ScoreNets(torch.autograd.Function):
    """EdgeScoreNetwork_dense + NodeScoreNetwork_dense on given node features / adjacency channels (the two networks
    without the SDE wrapper): raw pair scalar S [P] and node outputs [N, nout] before the masks."""

    @staticmethod
    def forward(ctx, cfg, x2, AC, flags, *T):
        dev, N, F = x2.device, cfg.N, x2.size(1)
        TE, TN = T[:cfg.n_edge_tensors], T[cfg.n_edge_tensors:]
        XC = _empty(N, F + 64, device=dev)
        XC[:, :F].copy_(x2)
        se = edge_forward(cfg, XC[:, :F], AC, flags, cfg.chans, cfg.offs, TE)
        sn = node_forward(cfg, XC, F, AC, TN)
        S = _empty(cfg.P, 1, device=dev)
        hip.gemm_ex(se.G2, TE[-2], S, bias=TE[-1])
        ctx.cfg, ctx.se, ctx.sn, ctx.T, ctx.keep = cfg, se, sn, T, (XC, AC, flags, F)
        return S.view(-1), sn.OUT[:, :sn.nout].contiguous()

    @staticmethod
    def backward(ctx, gS, gO):
        cfg, se, sn, T = ctx.cfg, ctx.se, ctx.sn, ctx.T
        XC, AC, flags, F = ctx.keep
        TE, TN = T[:cfg.n_edge_tensors], T[cfg.n_edge_tensors:]
        gS = gS.contiguous()
        sg = torch.sigmoid(se.ZG2)
        gZG2 = (gS[:, None] * TE[-2].detach().view(1, -1) * sg * (1 + se.ZG2 * (1 - sg))).contiguous()
        gOUT = torch.zeros(cfg.N, XP_LD, device=gO.device)
        gOUT[:, :sn.nout] = gO
        GN, gXC = node_backward(cfg, sn, XC, F, AC, TN, gOUT)
        GE, _ = edge_backward(cfg, se, AC, flags, cfg.chans, cfg.offs, TE, gS, gZG2, gXC[:, :F], True)
        return (None, gXC[:, :F].contiguous(), None, None) + tuple(GE) + tuple(GN)


def score_networks(edge, node, x, adj, flags):
    """(score_edge [B,N,N], score_node [B,N,nout]) = (edge(x, adj, flags), node(x, adj, flags)) of
    invariant_scorenetwork_dense.py:74-93,118-131 through the fused kernels, for padded inputs (every molecule is given
    N rows; `flags` masks).  The adjacency carries no gradient on this path."""
    B, N, F = x.shape
    dev = x.device
    cfg = types.SimpleNamespace(N=B * N, P=B * N * N, B=B, n_max=N)
    cfg.mol_ptr = (torch.arange(B + 1, device=dev) * N).to(torch.int32)
    cfg.pair_ptr = (torch.arange(B + 1, device=dev) * N * N).to(torch.int32)
    cfg.chans, cfg.offs = edge_net_shape(edge)
    TE, TN = edge_net_tensors(edge), node_net_tensors(node)
    cfg.n_edge_tensors = len(TE)
    adj = adj.detach()
    AC = torch.zeros(B * N * N, AC_LD, device=dev)
    AC[:, 0] = adj.reshape(-1)
    AC[:, 1] = torch.bmm(adj, adj).reshape(-1)                      # pow_tensor(adj, 2)
    S, O = _ScoreNets.apply(cfg, x.reshape(B * N, F).float(), AC, flags.reshape(-1).float().contiguous(), *(TE + TN))
    eye = torch.eye(N, device=dev)
    fm = flags[:, :, None] * flags[:, None, :]
    return S.view(B, N, N) * (1 - eye) * fm, O.view(B, N, -1) * flags[:, :, None]
