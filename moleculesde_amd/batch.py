"""Minimal mini-batch container with PyG 2.0.x collate rules.

The reference's models read only `.x .edge_index .edge_attr .positions .extended_edge_index
.batch .num_graphs` from a PyG `Batch` (SDE_model_2D_to_3D.py:307-321,
SDE_model_3D_to_2D_node_adj_dense.py:109-131).  This container provides exactly those, so the
model classes work on it or on a real PyG Batch (duck typing).
"""
import numpy as np
import torch


class MolData:
    """One molecule: x [n,9] (or [n]) int64, edge_index [2,e], edge_attr [e,3], positions [n,3]."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_nodes(self):
        return self.x.size(0)

    def keys(self):
        return [k for k, v in self.__dict__.items() if isinstance(v, torch.Tensor)]


def extend_graph_index(edge_index, num_nodes):
    """All ordered pairs within <= 4 bonds, no self loops, sorted row-major.

    Host restatement of Geom3D/datasets/dataset_3D.py:12-35 (A ∪ A², then (·) ∪ (·)², self loops
    removed, coalesced) as boolean matrix powers on the <= ~30-atom molecule.
    """
    n = int(num_nodes)
    A = np.zeros((n, n), dtype=bool)
    ei = edge_index.cpu().numpy()
    A[ei[0], ei[1]] = True
    for _ in range(2):
        P = (A.astype(np.int32) @ A.astype(np.int32)) > 0
        np.fill_diagonal(P, False)
        A = A | P
    r, c = np.nonzero(A)  # row-major sorted
    return torch.from_numpy(np.stack([r, c]).astype(np.int64))


class Batch:
    """Disjoint union of molecules.  Attributes whose name contains 'index' are concatenated along
    the last dim and offset by the running node count; everything else concatenates along dim 0;
    `batch` holds the molecule id per atom; `num_graphs` = number of molecules."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @staticmethod
    def from_data_list(data_list):
        keys = data_list[0].keys()
        out = {k: [] for k in keys}
        batch = []
        offset = 0
        for gid, d in enumerate(data_list):
            n = d.num_nodes
            for k in keys:
                v = getattr(d, k)
                out[k].append(v + offset if "index" in k else v)
            batch.append(torch.full((n,), gid, dtype=torch.long))
            offset += n
        b = Batch()
        for k in keys:
            setattr(b, k, torch.cat(out[k], dim=-1 if "index" in k else 0))
        b.batch = torch.cat(batch)
        b.num_graphs = len(data_list)
        return b

    def tensor_keys(self):
        return [k for k, v in self.__dict__.items() if isinstance(v, torch.Tensor)]

    def to(self, device, non_blocking=False):
        for k in self.tensor_keys():
            setattr(self, k, getattr(self, k).to(device, non_blocking=non_blocking))
        # any cached device-side graph plan is tied to the old tensors
        self.__dict__.pop("_msde_plan", None)
        return self

    def clone(self):
        b = Batch()
        for k, v in self.__dict__.items():
            if k == "_msde_plan":
                continue
            setattr(b, k, v.clone() if isinstance(v, torch.Tensor) else v)
        return b

    @property
    def num_nodes(self):
        return self.x.size(0)
