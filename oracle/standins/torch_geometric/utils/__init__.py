"""Stand-in utils: degree, softmax, to_dense_batch, to_dense_adj, add_self_loops (App. A.5/A.6)."""
import torch


def degree(index, num_nodes=None, dtype=None):
    N = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros((N,), dtype=dtype, device=index.device)
    return out.scatter_add_(0, index, torch.ones((index.size(0),), dtype=out.dtype, device=index.device))


def softmax(src, index, ptr=None, num_nodes=None, dim=0):
    N = int(index.max()) + 1 if num_nodes is None else num_nodes
    shape = [N] + list(src.shape[1:])
    idx = index.view([-1] + [1] * (src.dim() - 1)).expand_as(src)
    src_max = torch.full(shape, float("-inf"), dtype=src.dtype, device=src.device)
    src_max = src_max.scatter_reduce(0, idx, src.detach(), reduce="amax", include_self=True)
    out = (src - src_max.index_select(0, index)).exp()
    out_sum = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add(0, idx, out)
    return out / (out_sum.index_select(0, index) + 1e-16)


def add_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    N = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    loop = torch.arange(N, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index, torch.stack([loop, loop])], dim=1), edge_attr


def remove_self_loops(edge_index, edge_attr=None):
    m = edge_index[0] != edge_index[1]
    return edge_index[:, m], (None if edge_attr is None else edge_attr[m])


def to_dense_batch(x, batch=None, fill_value=0.0, max_num_nodes=None, batch_size=None):
    if batch is None:
        batch = x.new_zeros(x.size(0), dtype=torch.long)
    B = int(batch.max()) + 1 if batch_size is None else batch_size
    num_nodes = torch.zeros(B, dtype=torch.long, device=x.device).scatter_add_(0, batch, torch.ones_like(batch))
    cum = torch.cat([num_nodes.new_zeros(1), num_nodes.cumsum(0)])
    if max_num_nodes is None:
        max_num_nodes = int(num_nodes.max())
    idx = torch.arange(batch.size(0), device=x.device) - cum[batch] + batch * max_num_nodes
    size = [B * max_num_nodes] + list(x.shape[1:])
    out = x.new_full(size, fill_value)
    out[idx] = x
    out = out.view([B, max_num_nodes] + list(x.shape[1:]))
    mask = torch.zeros(B * max_num_nodes, dtype=torch.bool, device=x.device)
    mask[idx] = True
    return out, mask.view(B, max_num_nodes)


def to_dense_adj(edge_index, batch=None, edge_attr=None, max_num_nodes=None):
    if batch is None:
        batch = edge_index.new_zeros(int(edge_index.max()) + 1)
    B = int(batch.max()) + 1
    num_nodes = torch.zeros(B, dtype=torch.long, device=batch.device).scatter_add_(0, batch, torch.ones_like(batch))
    cum = torch.cat([num_nodes.new_zeros(1), num_nodes.cumsum(0)])
    idx0 = batch[edge_index[0]]
    idx1 = edge_index[0] - cum[batch][edge_index[0]]
    idx2 = edge_index[1] - cum[batch][edge_index[1]]
    if max_num_nodes is None:
        max_num_nodes = int(num_nodes.max())
    if edge_attr is None:
        edge_attr = torch.ones(idx0.numel(), device=edge_index.device)
    size = [B, max_num_nodes, max_num_nodes] + list(edge_attr.shape[1:])
    adj = torch.zeros(size, dtype=edge_attr.dtype, device=edge_index.device)
    flat = idx0 * max_num_nodes * max_num_nodes + idx1 * max_num_nodes + idx2
    adj = adj.view([-1] + list(edge_attr.shape[1:]))
    adj.index_add_(0, flat, edge_attr)
    return adj.view(size)


def subgraph(*a, **k):
    raise NotImplementedError


def to_networkx(*a, **k):
    raise NotImplementedError


def dense_to_sparse(*a, **k):
    raise NotImplementedError
