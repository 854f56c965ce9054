"""Canonical synthetic PCQM4Mv2-shaped / QM9-shaped / MD17-shaped batches.

No dataset exists in the build or GPU containers, so bench.py and the parity tests use this
generator.  Recipe (seeded, numpy Generator):
  n = clip(round(N(14.1, 2.9)), 2, 20) heavy atoms; bonds = random tree (parent in [i-3, i)) +
  Poisson(1.5) ring closures, both directions stored consecutively (dataset_utils.py:144-151);
  x[:,0] in {5,6,7,8,15,16}, other OGB columns uniform; positions = 1.5 A random walk along the
  tree, centred (dataset_3D.py:120-122); extended edges = pairs within <= 4 bonds
  (dataset_3D.py:12-35).
"""
import numpy as np
import torch

from .batch import Batch, MolData, extend_graph_index

ATOM_DIMS = [119, 4, 12, 12, 10, 6, 6, 2, 2]
BOND_DIMS_USED = [4, 3, 2]


def _tree_walk_positions(rng, n, parent, step=1.5):
    pos = np.zeros((n, 3), dtype=np.float64)
    for i in range(1, n):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d) + 1e-12
        pos[i] = pos[parent[i]] + step * d
    pos -= pos.mean(axis=0, keepdims=True)
    return pos.astype(np.float32)


def make_molecule(rng, n=None, with_h=False):
    if n is None:
        n = int(np.clip(np.rint(rng.normal(14.1, 2.9)), 2, 20))
    parent = np.zeros(n, dtype=np.int64)
    pairs = []
    for i in range(1, n):
        parent[i] = rng.integers(max(0, i - 3), i)
        pairs.append((int(parent[i]), i))
    have = set(pairs)
    for _ in range(int(rng.poisson(1.5))):
        if n < 3:
            break
        a, b = rng.integers(0, n, size=2)
        a, b = int(min(a, b)), int(max(a, b))
        if a != b and (a, b) not in have:
            have.add((a, b))
            pairs.append((a, b))
    src, dst, attr = [], [], []
    for a, b in pairs:
        ea = [int(rng.integers(0, d)) for d in BOND_DIMS_USED]
        src += [a, b]
        dst += [b, a]
        attr += [ea, ea]
    x = np.zeros((n, 9), dtype=np.int64)
    x[:, 0] = rng.choice([5, 6, 7, 8, 15, 16], size=n)
    if with_h:
        x[rng.random(n) < 0.5, 0] = 0
    for k in range(1, 9):
        x[:, k] = rng.integers(0, ATOM_DIMS[k], size=n)
    edge_index = torch.tensor([src, dst], dtype=torch.long).reshape(2, -1)
    d = MolData(
        x=torch.from_numpy(x),
        edge_index=edge_index,
        edge_attr=torch.tensor(attr, dtype=torch.long).reshape(-1, 3),
        positions=torch.from_numpy(_tree_walk_positions(rng, n, parent)),
    )
    d.extended_edge_index = extend_graph_index(edge_index, n)
    return d


def make_batch(num_graphs=256, seed=0, sizes=None):
    """PCQM4Mv2-shaped batch (config 2/3 of BASELINE.json)."""
    rng = np.random.default_rng(seed)
    mols = [make_molecule(rng, None if sizes is None else int(sizes[i])) for i in range(num_graphs)]
    return Batch.from_data_list(mols)


def make_qm9_batch(num_graphs=32, seed=0):
    """QM9-shaped batch (config 1): n ~ 18 atoms incl. H, <= 29; 1-D z in `x`."""
    rng = np.random.default_rng(seed)
    mols = []
    for _ in range(num_graphs):
        n = int(np.clip(np.rint(rng.normal(18.0, 3.0)), 3, 29))
        m = make_molecule(rng, n, with_h=True)
        m.x = m.x[:, 0].contiguous()
        mols.append(m)
    return Batch.from_data_list(mols)


def make_md17_batch(num_graphs=1, seed=0, n_atoms=21):
    """MD17-aspirin-shaped batch (config 5): 21 atoms, 1-D z."""
    rng = np.random.default_rng(seed)
    mols = []
    for _ in range(num_graphs):
        m = make_molecule(rng, n_atoms, with_h=True)
        m.x = m.x[:, 0].contiguous()
        mols.append(m)
    return Batch.from_data_list(mols)


def batch_stats(b, cutoff=10.0):
    """N, E_b, E_e, E_r (radius edges, strict < cutoff, no self loops), sum n^2, N_max."""
    N = b.x.size(0)
    pos = b.positions.detach().cpu().double()
    bid = b.batch.cpu()
    counts = torch.bincount(bid, minlength=b.num_graphs)
    E_r = 0
    start = 0
    for c in counts.tolist():
        p = pos[start:start + c]
        d2 = ((p[:, None] - p[None]) ** 2).sum(-1)
        E_r += int((d2 < cutoff * cutoff).sum()) - c
        start += c
    return {
        "N": int(N), "B": int(b.num_graphs), "E_b": int(b.edge_index.size(1)),
        "E_e": int(b.extended_edge_index.size(1)) if hasattr(b, "extended_edge_index") else 0,
        "E_r": int(E_r), "sum_n2": int((counts.double() ** 2).sum()), "N_max": int(counts.max()),
    }
