"""Stand-in torch_geometric.nn: MessagePassing, TransformerConv, radius_graph, pools. TEST-ONLY."""
import torch
from torch_scatter import scatter
from .conv import MessagePassing, TransformerConv  # noqa: F401


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow="source_to_target"):
    """App. A.1: for every target i, sources j in the same molecule with |xi-xj|^2 < r^2 (strict),
    j != i, at most max_num_neighbors (first hits in index order); returns [source; target]
    grouped by target."""
    assert flow == "source_to_target"
    N = x.size(0)
    if batch is None:
        batch = x.new_zeros(N, dtype=torch.long)
    d2 = ((x.unsqueeze(1) - x.unsqueeze(0)) ** 2).sum(-1)  # [target, source]
    ok = (d2 < r * r) & (batch.unsqueeze(1) == batch.unsqueeze(0))
    if not loop:
        ok = ok & ~torch.eye(N, dtype=torch.bool)
    rank = ok.long().cumsum(1)
    ok = ok & (rank <= max_num_neighbors)
    tgt, src = ok.nonzero(as_tuple=True)
    return torch.stack([src, tgt], dim=0)


def global_add_pool(x, batch, size=None):
    return scatter(x, batch, dim=0, dim_size=size, reduce="sum")


def global_mean_pool(x, batch, size=None):
    return scatter(x, batch, dim=0, dim_size=size, reduce="mean")


def global_max_pool(x, batch, size=None):
    raise NotImplementedError


class GlobalAttention(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError


class Set2Set(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError
