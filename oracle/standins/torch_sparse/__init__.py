"""TEST-ONLY stand-in for torch_sparse. Not product code."""
import torch


class SparseTensor:  # placeholder: only used in isinstance checks / annotations
    pass


def coalesce(index, value, m, n, op="add"):
    key = index[0] * n + index[1]
    uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
    out_index = torch.stack([uniq // n, uniq % n], dim=0)
    if value is None:
        return out_index, None
    out_val = torch.zeros((uniq.numel(),) + tuple(value.shape[1:]), dtype=value.dtype)
    out_val.index_add_(0, inv, value)
    return out_index, out_val


def spspmm(indexA, valueA, indexB, valueB, m, k, n, coalesced=False):
    A = torch.zeros(m, k, dtype=valueA.dtype)
    A.index_put_((indexA[0], indexA[1]), valueA, accumulate=True)
    B = torch.zeros(k, n, dtype=valueB.dtype)
    B.index_put_((indexB[0], indexB[1]), valueB, accumulate=True)
    # structural product: an entry exists wherever some path exists
    S = ((A != 0).float() @ (B != 0).float()) > 0
    C = A @ B
    idx = S.nonzero(as_tuple=False).t().contiguous()
    return idx, C[idx[0], idx[1]]
