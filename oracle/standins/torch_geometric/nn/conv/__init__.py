"""Stand-in MessagePassing + TransformerConv. TEST-ONLY."""
import inspect
import math

import torch
import torch.nn.functional as F
from torch_scatter import scatter
from torch_geometric.utils import softmax


class MessagePassing(torch.nn.Module):
    _special = {"index", "ptr", "size_i", "size_j", "edge_index", "dim_size"}

    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
        super().__init__()
        self.aggr = aggr
        self.flow = flow
        self.node_dim = node_dim
        self._msg_params = list(inspect.signature(self.message).parameters.keys())

    def propagate(self, edge_index, size=None, **kwargs):
        assert self.flow == "source_to_target"
        j, i = edge_index[0], edge_index[1]
        # number of target nodes
        N = None
        for v in kwargs.values():
            if isinstance(v, (tuple, list)) and isinstance(v[0], torch.Tensor):
                N = v[1].size(self.node_dim)
                break
            if isinstance(v, torch.Tensor) and v.dim() >= 2 and N is None:
                N = v.size(self.node_dim)
        if "x" in kwargs:
            xx = kwargs["x"]
            N = (xx[1] if isinstance(xx, (tuple, list)) else xx).size(self.node_dim)
        args = {}
        for name in self._msg_params:
            if name not in self._special and (name.endswith("_j") or name.endswith("_i")):
                base = kwargs[name[:-2]]
                if isinstance(base, (tuple, list)):
                    base = base[0] if name.endswith("_j") else base[1]
                idx = j if name.endswith("_j") else i
                args[name] = base.index_select(self.node_dim, idx)
            elif name == "index":
                args[name] = i
            elif name == "ptr":
                args[name] = None
            elif name == "size_i":
                args[name] = N
            elif name == "edge_index":
                args[name] = edge_index
            else:
                args[name] = kwargs[name]
        msg = self.message(**args)
        reduce = {"add": "sum", "sum": "sum", "mean": "mean"}[self.aggr]
        out = scatter(msg, i, dim=self.node_dim, dim_size=N, reduce=reduce)
        return self.update(out)

    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out


class TransformerConv(MessagePassing):
    def __init__(self, in_channels, out_channels, heads=1, concat=True, beta=False,
                 dropout=0.0, edge_dim=None, bias=True, root_weight=True, **kwargs):
        super().__init__(aggr="add", node_dim=0)
        assert concat and not beta and root_weight and bias
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.dropout = dropout
        self.lin_key = torch.nn.Linear(in_channels, heads * out_channels)
        self.lin_query = torch.nn.Linear(in_channels, heads * out_channels)
        self.lin_value = torch.nn.Linear(in_channels, heads * out_channels)
        self.lin_edge = torch.nn.Linear(edge_dim, heads * out_channels, bias=False)
        self.lin_skip = torch.nn.Linear(in_channels, heads * out_channels, bias=True)

    def forward(self, x, edge_index, edge_attr=None):
        out = self.propagate(edge_index, x=(x, x), edge_attr=edge_attr)
        out = out.view(-1, self.heads * self.out_channels)
        return out + self.lin_skip(x)

    def message(self, x_i, x_j, edge_attr, index, ptr, size_i):
        H, C = self.heads, self.out_channels
        query = self.lin_query(x_i).view(-1, H, C)
        key = self.lin_key(x_j).view(-1, H, C)
        e = self.lin_edge(edge_attr).view(-1, H, C)
        key = key + e
        alpha = (query * key).sum(dim=-1) / math.sqrt(C)
        alpha = softmax(alpha, index, ptr, size_i)
        alpha = F.dropout(alpha, p=self.dropout, training=self.training)
        out = self.lin_value(x_j).view(-1, H, C) + e
        return out * alpha.view(-1, H, 1)
