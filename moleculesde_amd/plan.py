"""Per-batch graph plans: everything index-shaped that the kernels need, built ONCE per mini-batch
(at collation time on the host, next to PyG-style collate, or lazily from a device batch).

The reference rebuilds nothing but the radius graph per step (schnet.py:91); its bond and extended
edge lists come from the data loader (dataset_3D.py:12-35).  Accordingly the bond / extended CSR
plans and embedding row lists live here, while the radius CSR is produced on the device in every
SchNet forward (hip.radius_plan).
"""
import types

import torch

from . import hip

ATOM_FEATURE_DIMS = [119, 4, 12, 12, 10, 6, 6, 2, 2]   # ogb 1.2.1; configurable in GNN
BOND_FEATURE_DIMS = [5, 6, 2]


def _offsets(dims):
    off = [0]
    for d in dims[:-1]:
        off.append(off[-1] + d)
    return off


def _row_lists(codes, R):
    """CSR of item ids per table row: codes [M,K] (already offset) -> (list_ptr [R+1], list_items)."""
    M, K = codes.shape
    flat = codes.reshape(-1).long()
    item = torch.arange(M, device=codes.device).repeat_interleave(K)
    srow, order = torch.sort(flat, stable=True)
    list_ptr = torch.searchsorted(srow, torch.arange(R + 1, device=codes.device)).to(torch.int32)
    return list_ptr, item[order].to(torch.int32)


def build_plan(data, atom_dims=None, bond_dims=None, max_nbr=32, with_ext=True):
    """Build the plan from the batch tensors wherever they live (host preferred)."""
    atom_dims = atom_dims or ATOM_FEATURE_DIMS
    bond_dims = bond_dims or BOND_FEATURE_DIMS
    pl = types.SimpleNamespace()
    x = data.x
    N = x.size(0)
    dev = x.device
    pl.N = N
    batch = data.batch
    B = int(data.num_graphs) if hasattr(data, "num_graphs") else int(batch.max()) + 1
    pl.B = B
    counts = torch.bincount(batch, minlength=B)
    pl.mol_ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).to(torch.int32)
    pl.batch_i32 = batch.to(torch.int32)
    c = counts.cpu()
    pl.E_r_cap = int((c * torch.clamp(c - 1, max=max_nbr)).sum())
    pl.N_max = int(c.max()) if B > 0 else 0
    pl.P2_cap = int((c * (c - 1) // 2).sum())      # unordered atom pairs (hip.pair_plan)
    pl.max_nbr = max_nbr
    # 2D atom codes (9 OGB columns) or 1-D z
    if x.dim() == 2:
        aoff = torch.tensor(_offsets(atom_dims), device=dev)
        pl.atom_codes = (x.long() + aoff[None, : x.size(1)]).to(torch.int32).contiguous()
        pl.atom_R = sum(atom_dims)
        pl.atom_list_ptr, pl.atom_list_nodes = _row_lists(pl.atom_codes, pl.atom_R)
        z = x[:, 0]
    else:
        z = x
    pl.z_codes = z.to(torch.int32).view(-1, 1).contiguous()
    pl.z_list = None  # built lazily per node_class (SchNet.embedding rows)
    # bond graph
    if hasattr(data, "edge_index") and data.edge_index is not None:
        pl.bond = hip.build_csr(data.edge_index, N)
        if hasattr(data, "edge_attr") and data.edge_attr is not None and data.edge_attr.dim() == 2:
            boff = torch.tensor(_offsets(bond_dims), device=dev)
            ea = data.edge_attr.long()[pl.bond.perm_t]
            pl.bond_codes = (ea + boff[None, : ea.size(1)]).to(torch.int32).contiguous()
            pl.bond_R = sum(bond_dims)
            pl.bond_type = ea[:, 0].to(torch.float32)  # canonical order; 3D->2D adjacency values
    if with_ext and hasattr(data, "extended_edge_index") and data.extended_edge_index is not None:
        pl.ext = hip.build_csr(data.extended_edge_index, N)
    return pl


def plan_to(pl, device):
    for k, v in list(vars(pl).items()):
        if isinstance(v, torch.Tensor):
            setattr(pl, k, v.to(device))
        elif isinstance(v, hip.CsrPlan):
            v.to(device)
        elif isinstance(v, types.SimpleNamespace):
            for kk, vv in list(vars(v).items()):
                if isinstance(vv, torch.Tensor):
                    setattr(v, kk, vv.to(device))
    return pl


def prepare_batch(data, device=None, **kw):
    """Collate-time entry point: build the plan on the host, then move batch + plan to the device."""
    pl = build_plan(data, **kw)
    if getattr(data, "edge_attr", None) is not None and data.edge_attr.dim() == 2:
        dense_plan(pl, data)
    if device is not None:
        data.to(device)
        plan_to(pl, device)
    data._msde_plan = pl
    return data


def get_plan(data):
    """Plan of a batch, built lazily (one host sync) when the driver did not call prepare_batch."""
    pl = getattr(data, "_msde_plan", None)
    if pl is None or pl.mol_ptr.device != data.x.device:
        pl = build_plan(data)
        try:
            data._msde_plan = pl
        except Exception:
            pass
    return pl


def dense_plan(pl, data):
    """Padded (dense) layout of the batch for the 3D->2D head, built once per batch from data only
    (SDE_model_3D_to_2D_node_adj_dense.py:121-134): N_max, slot maps, adj [B,Nm,Nm] with bond type + 1,
    flags (atoms with a non-zero adjacency row), padded atom classes z."""
    dn = getattr(pl, "dense", None)
    if dn is not None:
        return dn
    dn = types.SimpleNamespace()
    dev = data.x.device
    B, N, Nm = pl.B, pl.N, pl.N_max
    batch = data.batch
    mol_ptr = pl.mol_ptr.long()
    local = torch.arange(N, device=dev) - mol_ptr[batch]
    slot = batch * Nm + local
    pad_idx = torch.full((B * Nm,), -1, dtype=torch.int32, device=dev)
    pad_idx[slot] = torch.arange(N, device=dev, dtype=torch.int32)
    dn.N_max, dn.pad_idx, dn.node_slot = Nm, pad_idx, slot.to(torch.int32)
    ei = data.edge_index
    val = data.edge_attr[:, 0].float() + 1                               # bond type + 1 (:121)
    b = batch[ei[0]]
    flat = b * Nm * Nm + (ei[0] - mol_ptr[b]) * Nm + (ei[1] - mol_ptr[b])
    adj = torch.zeros(B * Nm * Nm, dtype=torch.float32, device=dev).index_add_(0, flat, val)
    dn.adj = adj.view(B, Nm, Nm)
    dn.flags = torch.abs(dn.adj).sum(-1).gt(1e-5).to(torch.float32)      # node_flags (:523-529)
    z = torch.zeros(B * Nm, dtype=torch.long, device=dev)
    z[slot] = data.x[:, 0].long() if data.x.dim() == 2 else data.x.long()
    dn.z = z.view(B, Nm)
    # ragged pair layout of the fused head kernels (csrc/dense_head.hip): molecule b owns pair rows
    # pair_ptr[b] + i * n_b + j
    cnt = (mol_ptr[1:] - mol_ptr[:-1])
    sq = (cnt * cnt).cumsum(0)
    dn.pair_ptr = torch.cat([sq.new_zeros(1), sq]).to(torch.int32)
    dn.P = int(sq[-1]) if B > 0 else 0
    pl.dense = dn
    return dn


def z_lists(pl, node_class):
    """Row lists for SchNet's nn.Embedding(node_class, H) backward."""
    if pl.z_list is None or pl.z_list[0] != node_class:
        ptr, nodes = _row_lists(pl.z_codes, node_class)
        pl.z_list = (node_class, ptr, nodes)
    return pl.z_list[1], pl.z_list[2]
