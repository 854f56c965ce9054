"""EquivariantScoreNetwork on the one-workgroup-per-molecule kernels (csrc/escore_mol.hip).

`forward(net, ...)` evaluates the whole network of equivariant_scorenetwork.py:121-169 (4 GATLayers, 2 basis MLPs, frame
mix, mean over in-edges) with ONE launch; used by `SDEModel2Dto3D_02.forward / get_score` (SDE_model_2D_to_3D.py:386-391,
:393-445) whenever `supported(net, pl)`.  The operator path of geom3d/sde_2d_to_3d.py stays as the cross-check and for the
shapes the kernel does not take.
"""
import ctypes

import torch

from . import _lib, hip

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
    """The 52 tensors of the kernel's parameter table: 11 per GAT layer, then 4 per basis MLP (struct EsW).  q|k|v|skip
    weights / biases are the (free, once FlatAdam laid them out back to back) concatenations the operator path uses."""
    out = []
    for g in (g for blk in net.gnn_layers for g in blk):
        Ws, bs = g.MHA.fusion_sets()
        out += [hip.cat_params(Ws), hip.cat_params(bs), g.MHA.lin_edge.weight, g.norm1.weight, g.norm1.bias, g.FFN[0].weight,
                g.FFN[0].bias, g.FFN[3].weight, g.FFN[3].bias, g.norm2.weight, g.norm2.bias]
    for m in net.basis_mlp_modules:
        out += [m[0].weight, m[0].bias, m[2].weight, m[2].bias]
    return out


_TABLES = {}          # id(net) -> (addresses, device int64 tensor of the 52 pointers, the tensors themselves)


def _pointer_table(net, tensors):
    """Device array of the 52 parameter addresses.  Rebuilt only when an address changes (FlatAdam keeps them fixed), never
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
              int(pl.B), hip._p(ep.rowptr), hip._p(ep.src), hip._p(ep.dst), ep.N, ep.E, 32, 8, 128, p_att, p_ffn,
              int(seed0) & 0xFFFFFFFFFFFFFFFF, hip._p(seed_dev), eps1, eps2, hip._p(out), hip._p(None), hip._stream())
    return out
