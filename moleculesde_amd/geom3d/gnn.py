"""GIN 2D encoder on HIP kernels.  Mirrors the API of Geom3D/models/molecule_gnn_model.py:132-197
(`GNN(num_layer, emb_dim, JK, drop_ratio, gnn_type)`, `forward(x, edge_index, edge_attr)` or
`forward(data)`), with identical state_dict keys, so reference checkpoints load unchanged."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip, plan as _plan
from . import nn as _nn


import os as _os
FUSE_GIN_LAYER = True     # BatchNorm folded into the GIN products (False: separate BatchNorm launches, the cross-check)
FUSE_GIN_APPLY = True     # ... and its apply into the next layer's aggregation


class GINConv(nn.Module):
    """molecule_gnn_model.py:13-32.  The bond-embedding sum, the gather of x_j, the ReLU and the
    per-target segmented sum run as one kernel (hip.gin_aggregate)."""

    def __init__(self, emb_dim, bond_dims):
        super().__init__()
        bn = _nn.BatchNorm1d(2 * emb_dim)
        bn.fuse_relu = True                      # mlp[2] (ReLU) is fused into the BatchNorm kernel
        self.mlp = nn.Sequential(_nn.Linear(emb_dim, 2 * emb_dim), bn, nn.Identity(), _nn.Linear(2 * emb_dim, emb_dim))
        self.eps = nn.Parameter(torch.Tensor([0]))
        self.bond_encoder = _nn.EmbeddingList(bond_dims, emb_dim, "bond_embedding_list")

    def forward(self, x, bond_plan, bond_codes, outer_bn=None, link=None, defer_apply=False):
        """outer_bn: the layer's BatchNorm of GNN.batch_norms (molecule_gnn_model.py:176-182).  Given in training mode,
        the whole Linear -> BatchNorm -> ReLU -> Linear -> BatchNorm (-> ReLU) chain runs as hip.gin_mlp_bn (statistics
        and normalisation folded into the products) and the result already includes outer_bn."""
        agg = hip.gin_aggregate(x, self.bond_encoder.table(), self.eps, bond_plan, bond_codes, link)
        if outer_bn is None:
            return self.mlp(agg)
        for bn in (self.mlp[1], outer_bn):
            _nn.count_batch(bn)
        return hip.gin_mlp_bn(agg, self.mlp[0], self.mlp[1], self.mlp[3], outer_bn, outer_bn.fuse_relu, defer_apply)


class GNN(nn.Module):
    def __init__(self, num_layer, emb_dim, JK="last", drop_ratio=0, gnn_type="gin", atom_feature_dims=None,
                 bond_feature_dims=None):
        super().__init__()
        self.num_layer, self.drop_ratio, self.JK = num_layer, drop_ratio, JK
        if self.num_layer < 2:
            raise ValueError("Number of GNN layers must be greater than 1.")
        if str(gnn_type).upper() != "GIN":
            raise NotImplementedError(f"gnn_type={gnn_type!r}: only GIN is on the MoleculeSDE hot path")
        self.atom_dims = list(atom_feature_dims or _plan.ATOM_FEATURE_DIMS)
        self.bond_dims = list(bond_feature_dims or _plan.BOND_FEATURE_DIMS)
        self.atom_encoder = _nn.EmbeddingList(self.atom_dims, emb_dim, "atom_embedding_list")
        self.gnns = nn.ModuleList([GINConv(emb_dim, self.bond_dims) for _ in range(num_layer)])
        self.batch_norms = nn.ModuleList([_nn.BatchNorm1d(emb_dim) for _ in range(num_layer)])
        for layer in range(num_layer - 1):
            self.batch_norms[layer].fuse_relu = True     # ReLU after every layer but the last (:178-182)

    def _find_plan(self, x, edge_index, edge_attr, data=None):
        pl = None
        if data is not None:
            pl = _plan.get_plan(data)
        if pl is None:
            pl = _nn.lookup_plan(edge_index) or _nn.lookup_plan(x)
        if pl is None or not hasattr(pl, "bond_codes"):
            # generic caller (no prepare_batch): build the bond plan from the raw tensors on the device
            import types
            d = types.SimpleNamespace(x=x, edge_index=edge_index, edge_attr=edge_attr,
                                      batch=torch.zeros(x.size(0), dtype=torch.long, device=x.device), num_graphs=1)
            pl = _plan.build_plan(d, atom_dims=self.atom_dims, bond_dims=self.bond_dims, with_ext=False)
        return pl

    def forward(self, *argv):
        if len(argv) == 3:
            x, edge_index, edge_attr = argv
            pl = self._find_plan(x, edge_index, edge_attr)
        elif len(argv) == 1:
            data = argv[0]
            x, edge_index, edge_attr = data.x, data.edge_index, data.edge_attr
            pl = self._find_plan(x, edge_index, edge_attr, data)
        else:
            raise ValueError("unmatched number of arguments.")

        h = hip.embedding_sum(self.atom_encoder.table(), pl.atom_codes, pl.atom_list_ptr, pl.atom_list_nodes)
        h_list = [h]
        fused = FUSE_GIN_LAYER and self.training and h.is_cuda and h.size(1) % 4 == 0 and 0 < h.size(0) <= hip.RS_MAX_ROWS
        # a layer output that only the next layer's aggregation reads (JK = last, no dropout) is never normalised by a launch
        # of its own: the aggregation kernel applies the BatchNorm while it gathers
        lazy_ok = fused and FUSE_GIN_APPLY and self.JK == "last" and self.drop_ratio == 0
        link = None
        for layer in range(self.num_layer):
            if fused and _nn.bn_fusable(self.gnns[layer].mlp[1]) and _nn.bn_fusable(self.batch_norms[layer]):
                defer = lazy_ok and layer + 1 < self.num_layer
                out = self.gnns[layer](h_list[layer], pl.bond, pl.bond_codes, self.batch_norms[layer], link, defer)
                h, link = out if defer else (out, None)
            else:
                if link is not None:           # the previous layer deferred its BatchNorm apply: do it now
                    hip.bn_apply_link(h_list[layer], link)
                    link = None
                h = self.gnns[layer](h_list[layer], pl.bond, pl.bond_codes)
                h = self.batch_norms[layer](h)             # ReLU fused for all but the last layer
            if self.drop_ratio > 0:
                h = F.dropout(h, self.drop_ratio, training=self.training)
            h_list.append(h)

        if self.JK == "concat":
            return torch.cat(h_list, dim=1)
        if self.JK == "last":
            return h_list[-1]
        if self.JK == "max":
            return torch.max(torch.stack(h_list, dim=0), dim=0)[0]
        if self.JK == "sum":
            return torch.sum(torch.stack(h_list, dim=0), dim=0)
        raise ValueError(self.JK)
