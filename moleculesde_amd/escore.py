"""EquivariantScoreNetwork on the one-workgroup-per-molecule kernels (csrc/escore_mol.hip).

`forward(net, ...)` evaluates the whole network of equivariant_scorenetwork.py:121-169 (4 GATLayers, 2 basis MLPs, frame
mix, mean over in-edges) with ONE launch; used by `SDEModel2Dto3D_02.forward / get_score` (SDE_model_2D_to_3D.py:386-391,
:393-445) whenever `supported(net, pl)`.  The operator path of geom3d/sde_2d_to_3d.py stays as the cross-check and for the
shapes the kernel does not take.
"""
import ctypes

import torch

from . import _lib, hip, slabs

NMAX = 32                 # ES_NMAX of csrc/escore_mol.hip


def supported(net, pl):
    """hidden 32, 8 heads, basis-MLP width 128, molecules of at most 32 atoms, one LayerNorm eps per position."""
    if net.hidden_dim != 32 or net.num_head != 8 or net.hidden_coff_dim != 128 or net.num_layers != 2 or net.num_convs != 2:
        return False
    if pl is None or getattr(pl, "N_max", NMAX + 1) > NMAX or getattr(pl, "mol_ptr", None) is None:
        return False
    layers = [g for blk in net.gnn_layers for g in blk]
    return (len({g.norm1.eps for g in layers}) == 1 and len({g.norm2.eps for g in layers}) == 1 and
            len({g.MHA.dropout for g in layers}) == 1 and len({g.FFN[2].p for g in layers}) == 1)


def param_tensors(net):
    """The 76 parameters of the kernel's table: 17 per GAT layer, then 4 per basis MLP (struct EsW, csrc/escore_mol.h)."""
    out = []
    for g in (g for blk in net.gnn_layers for g in blk):
        Ws, bs = g.MHA.fusion_sets()                 # [query, key, value, skip]
        out += list(Ws) + list(bs) + [g.MHA.lin_edge.weight, g.norm1.weight, g.norm1.bias, g.FFN[0].weight, g.FFN[0].bias,
                                      g.FFN[3].weight, g.FFN[3].bias, g.norm2.weight, g.norm2.bias]
    for m in net.basis_mlp_modules:
        out += [m[0].weight, m[0].bias, m[2].weight, m[2].bias]
    return out


_TABLES = {}          # id(net) -> (addresses, device int64 tensor of the 76 pointers, the tensors themselves)


def _pointer_table(net, tensors):
    """Device array of the 76 parameter addresses.  Rebuilt only when an address changes (FlatAdam keeps them fixed), never
    inside a hipGraph capture: an eager call of the same network must come first (the trainer's warm-up steps do)."""
    addrs = tuple(t.data_ptr() for t in tensors)
    hit = _TABLES.get(id(net))
    if hit is not None and hit[0] == addrs and hit[1].device == tensors[0].device:
        hit[2][:] = tensors                      # keep the views alive for as long as their addresses are in the table
        return hit[1]
    for t in tensors:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0):
            raise _lib.MsdeHipError("escore: parameters must be contiguous fp32 device tensors, 16-byte aligned")
    if torch.cuda.is_current_stream_capturing():
        raise _lib.MsdeHipError("escore: parameter table missing inside a capture (run one eager call first)")
    tab = torch.tensor(addrs, dtype=torch.int64).to(tensors[0].device)
    _TABLES[id(net)] = (addrs, tab, list(tensors))
    import weakref
    weakref.finalize(net, _TABLES.pop, id(net), None)
    return tab


def _cfg(net):
    g = net.gnn_layers[0][0]
    train = net.training
    return (float(g.MHA.dropout if train else 0.0), float(g.FFN[2].p if train else 0.0), float(g.norm1.eps), float(g.norm2.eps))


def forward_nograd(net, ep, pl, node_attr, edge_attr, basis, seed0, seed_dev):
    """Inference / no-autograd evaluation: returns the score [N, 3]."""
    tens = param_tensors(net)
    p_att, p_ffn, eps1, eps2 = _cfg(net)
    x0, ea = hip._f32(node_attr), edge_attr
    if not (ea.is_cuda and ea.dtype == torch.float32 and ea.stride(-1) == 1 and ea.stride(0) % 4 == 0 and ea.data_ptr() % 16 == 0):
        ea = hip._f32(ea)
    out = torch.empty(ep.N, 3, dtype=torch.float32, device=x0.device)
    _lib.call("msde_escore_mol_fwd", hip._p(_pointer_table(net, tens)), hip._p(x0), hip._p(ea), ea.stride(0), hip._p(basis), hip._p(pl.mol_ptr),
              int(pl.B), hip._p(ep.rowptr), hip._p(ep.src), hip._p(ep.dst), ep.N, ep.E, 32, 8, 128, int(pl.N_max), p_att, p_ffn,
              int(seed0) & 0xFFFFFFFFFFFFFFFF, hip._p(seed_dev), eps1, eps2, hip._p(out), hip._p(None), hip._stream())
    return out


def score_supported(model, pl):
    """msde_escore_mol_score: what `supported` asks of the score network (molecules of at most 32 atoms), and the
    coordinate branch in the reference's shape (input_mlp one Linear, project Linear -> SiLU -> Linear, no dropout)."""
    net = model.score_network
    if not supported(net, pl) or model.hidden_dim != 32:
        return False
    proj = model.project
    if len(proj.layers) != 2 or proj.activation_name != "silu" or proj.dropout:
        return False
    if hasattr(model, "input_mlp") and len(model.input_mlp.layers) != 1:
        return False
    return True


def score_param_tensors(model):
    """The 86 parameters of msde_escore_mol_score's table (include/msde_hip.h)."""
    has_dist = hasattr(model, "input_mlp")
    cf = model.coff_gaussian_fourier.W
    geo = [model.dist_gaussian_fourier.W if has_dist else cf, cf]
    if has_dist:
        geo += [model.input_mlp.layers[0].weight, model.input_mlp.layers[0].bias]
    else:
        geo += [model.coff_mlp.weight, model.coff_mlp.bias]          # placeholders: never read when has_dist = 0
    geo += [model.coff_mlp.weight, model.coff_mlp.bias, model.project.layers[0].weight, model.project.layers[0].bias,
            model.project.layers[1].weight, model.project.layers[1].bias]
    return param_tensors(model.score_network) + geo


def score_nograd(model, ep, pl, node_attr, edge_2D, pos):
    """SDEModel2Dto3D_0x.get_score's network output [N, 3] (before the division by -std): two launches (per-edge work wide,
    then one workgroup per molecule)."""
    net = model.score_network
    tens = score_param_tensors(model)
    _, _, eps1, eps2 = _cfg(net)
    x0, e2, pos = hip._f32(node_attr), edge_2D, hip._f32(pos)
    if not (e2.is_cuda and e2.dtype == torch.float32 and e2.stride(-1) == 1 and e2.stride(0) % 4 == 0 and e2.data_ptr() % 16 == 0):
        e2 = hip._f32(e2)
    out = torch.empty(ep.N, 3, dtype=torch.float32, device=x0.device)
    scratch = torch.empty(int(_lib.load().msde_escore_mol_score_scratch_floats(ep.E)), dtype=torch.float32, device=x0.device)
    _lib.call("msde_escore_mol_score", hip._p(_pointer_table(model, tens)), hip._p(x0), hip._p(pos), hip._p(e2), e2.stride(0),
              int(hasattr(model, "input_mlp")), hip._p(pl.mol_ptr), int(pl.B), hip._p(ep.rowptr), hip._p(ep.src), hip._p(ep.dst),
              ep.N, ep.E, 32, 8, 128, int(pl.N_max), eps1, eps2, hip._p(scratch), hip._p(out), hip._stream())
    return out


_LAYER_SHAPES = [(32, 32)] * 4 + [(32,)] * 4 + [(32, 32), (32,), (32,), (32, 32), (32,), (32, 32), (32,), (32,), (32,)]
_BASIS_SHAPES = [(128, 64), (128,), (3, 128), (3,)]


def _grad_views(gall):
    """Views of the summed slab [ES_SLAB] in the order of param_tensors (layout: csrc/escore_mol.h)."""
    out, off = [], 0
    for shapes, reps, pad in ((_LAYER_SHAPES, 4, 0), (_BASIS_SHAPES, 2, 1)):
        for _ in range(reps):
            for shp in shapes:
                n = 1
                for d in shp:
                    n *= d
                out.append(gall[off:off + n].view(shp))
                off += n
            off += pad
    assert off == gall.numel()
    return out


class _EScoreMol(torch.autograd.Function):
    """EquivariantScoreNetwork.forward as ONE autograd node: msde_escore_mol_fwd / msde_escore_mol_bwd, one workgroup per
    molecule each way.  Weight gradients leave the backward kernel as one slab per molecule and are summed by the batched slab
    reduction of the step (slabs._SLABS) or, outside a batch, right here."""

    @staticmethod
    def forward(ctx, node_attr, edge_attr, basis, net, ep, pl, seed0, seed_dev, *params):
        p_att, p_ffn, eps1, eps2 = _cfg(net)
        x0, ea = hip._f32(node_attr), edge_attr
        if not (ea.is_cuda and ea.dtype == torch.float32 and ea.stride(-1) == 1 and ea.stride(0) % 4 == 0 and ea.data_ptr() % 16 == 0):
            ea = hip._f32(ea)
        basis = hip._f32(basis)
        dev = x0.device
        N, E, B = ep.N, ep.E, int(pl.B)
        out = torch.empty(N, 3, dtype=torch.float32, device=dev)
        sv = torch.empty(int(_lib.load().msde_escore_mol_saved_floats(N)), dtype=torch.float32, device=dev)
        tab = _pointer_table(net, list(params))
        seed0 = int(seed0) & 0xFFFFFFFFFFFFFFFF
        _lib.call("msde_escore_mol_fwd", hip._p(tab), hip._p(x0), hip._p(ea), ea.stride(0), hip._p(basis), hip._p(pl.mol_ptr), B,
                  hip._p(ep.rowptr), hip._p(ep.src), hip._p(ep.dst), N, E, 32, 8, 128, int(pl.N_max), p_att, p_ffn, seed0,
                  hip._p(seed_dev), eps1, eps2, hip._p(out), hip._p(sv), hip._stream())
        ctx.save_for_backward(x0, ea, basis, sv, *params)
        ctx.cfg = (net, ep, pl, seed0, seed_dev, p_att, p_ffn, eps1, eps2, tab)
        ctx.deferrable = all(t.is_leaf for t in params)
        return out

    @staticmethod
    def backward(ctx, g):
        x0, ea, basis, sv = ctx.saved_tensors[:4]
        net, ep, pl, seed0, seed_dev, p_att, p_ffn, eps1, eps2, tab = ctx.cfg
        g = g if (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()) else hip._f32(g)
        dev = g.device
        N, E, B = ep.N, ep.E, int(pl.B)
        nslab = int(_lib.load().msde_escore_mol_slab_floats())
        g_x0 = torch.empty(N, 32, dtype=torch.float32, device=dev)
        g_ea = torch.empty(E, 32, dtype=torch.float32, device=dev)
        gall = torch.empty(nslab, dtype=torch.float32, device=dev)
        defer = slabs._SLABS.active and ctx.deferrable
        ws = slabs._SLABS.alloc(B * nslab, dev) if defer else torch.empty(B * nslab, dtype=torch.float32, device=dev)
        _lib.call("msde_escore_mol_bwd", hip._p(tab), hip._p(x0), hip._p(ea), ea.stride(0), hip._p(basis), hip._p(pl.mol_ptr), B,
                  hip._p(ep.rowptr), hip._p(ep.src), hip._p(ep.dst), hip._p(ep.rowptr_s), hip._p(ep.perm_s), N, E, 32, 8, 128,
                  int(pl.N_max), p_att, p_ffn, seed0, hip._p(seed_dev), eps1, eps2, hip._p(sv), hip._p(g), hip._p(g_x0), hip._p(g_ea), 32,
                  hip._p(ws), hip._stream())
        if defer:
            slabs._SLABS.add(ws.data_ptr(), B, nslab, gall, written=True)
        else:
            torch.sum(ws.view(B, nslab), dim=0, out=gall)
        return (g_x0, g_ea, None, None, None, None, None, None) + tuple(_grad_views(gall))


def forward(net, ep, pl, node_attr, edge_attr, basis, seed0, seed_dev):
    """The score [N, 3] of EquivariantScoreNetwork.forward, differentiable w.r.t. node_attr, edge_attr and all parameters."""
    if not torch.is_grad_enabled():
        return forward_nograd(net, ep, pl, node_attr, edge_attr, basis, seed0, seed_dev)
    return _EScoreMol.apply(node_attr, edge_attr, basis, net, ep, pl, seed0, seed_dev, *param_tensors(net))
