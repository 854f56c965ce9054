"""TEST-ONLY stand-in for torch_scatter. Not product code."""
import torch


def _expand_index(index, src, dim):
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() == 1 and src.dim() > 1:
        shape = [1] * src.dim()
        shape[dim] = -1
        index = index.view(shape).expand_as(src)
    return index, dim


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    index, dim = _expand_index(index, src, dim)
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    res = torch.zeros(shape, dtype=src.dtype, device=src.device)
    res = res.scatter_add(dim, index, src)
    if reduce in ("sum", "add"):
        return res
    if reduce == "mean":
        ones = torch.ones_like(src)
        cnt = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add(dim, index, ones)
        cnt = cnt.clamp(min=1)
        if src.is_floating_point():
            return res / cnt
        return torch.div(res, cnt, rounding_mode="floor")
    raise NotImplementedError(reduce)


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "sum")


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "mean")
