"""SchNet 3D encoder on HIP kernels.  Mirrors Geom3D/models/schnet.py:16-125: same constructor,
`forward(z, pos, batch=None, return_latent=False)` and state_dict keys (including the duplicated
`interactions.i.mlp.*` / `interactions.i.conv.nn.*` entries and the float64 `atomic_mass` buffer).

Device pipeline per forward:
  radius-graph CSR (count -> scan -> fill, no host sync)  ->  Gaussian smearing + cosine cutoff
  -> 6 x [ filter MLP, lin1, CFConv gather*filter segmented sum, lin2, ssp, lin, residual ]
  -> head -> per-molecule readout.
Under torch.no_grad() with num_filters == 128 the whole CFConv (smearing, filter MLP, cutoff,
gather, segmented sum) is ONE fp32-MFMA kernel (csrc/cfconv_fused.hip).
"""
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip, plan as _plan
from . import nn as _nn


FUSE_TAIL = True     # softplus / residual in GEMM epilogues (False: layer by layer, the cross-check)
# (Round 3's chained-product kernel -- lin2 -> ssp -> lin -> + residual -> the next lin1 as ONE launch on 16-row strips -- was
# 28.8 us against 30.0 us for the three launches and 2.843 vs 2.807 ms in the step: removed, numbers in DESIGN.md section 4.22.)


class GaussianSmearing(nn.Module):
    """Parameters of schnet.py:198-207; the expansion itself is fused into the edge kernels."""

    def __init__(self, start=0.0, stop=5.0, num_gaussians=50):
        super().__init__()
        offset = torch.linspace(start, stop, num_gaussians)
        self.coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
        self.num_gaussians = num_gaussians
        self.register_buffer("offset", offset)


class CFConv(nn.Module):
    def __init__(self, in_channels, out_channels, num_filters, nn_, cutoff):
        super().__init__()
        self.lin1 = _nn.Linear(in_channels, num_filters, bias=False)
        self.lin2 = _nn.Linear(num_filters, out_channels)
        self.nn = nn_
        self.cutoff = cutoff
        nn.init.xavier_uniform_(self.lin1.weight)
        nn.init.xavier_uniform_(self.lin2.weight)
        self.lin2.bias.data.fill_(0)


class InteractionBlock(nn.Module):
    def __init__(self, hidden_channels, num_gaussians, num_filters, cutoff):
        super().__init__()
        self.mlp = nn.Sequential(_nn.Linear(num_gaussians, num_filters), _nn.ShiftedSoftplus(),
                                 _nn.Linear(num_filters, num_filters))
        self.conv = CFConv(hidden_channels, hidden_channels, num_filters, self.mlp, cutoff)
        self.act = _nn.ShiftedSoftplus()
        self.lin = _nn.Linear(hidden_channels, hidden_channels)
        nn.init.xavier_uniform_(self.mlp[0].weight)
        self.mlp[0].bias.data.fill_(0)
        nn.init.xavier_uniform_(self.mlp[2].weight)   # mlp[2].bias keeps nn.Linear's default (App. B.1)
        nn.init.xavier_uniform_(self.lin.weight)
        self.lin.bias.data.fill_(0)


class SchNet(nn.Module):
    def __init__(self, hidden_channels=128, num_filters=128, num_interactions=6, num_gaussians=50, cutoff=10.0,
                 node_class=None, readout="mean", dipole=False, mean=None, std=None, atomref=None):
        super().__init__()
        assert readout in ["add", "sum", "mean"]
        if dipole or mean is not None or std is not None or atomref is not None:
            raise NotImplementedError("dipole / mean-std / atomref branches are never enabled on the MoleculeSDE path")
        self.hidden_channels, self.num_filters = hidden_channels, num_filters
        self.num_interactions, self.num_gaussians, self.cutoff = num_interactions, num_gaussians, cutoff
        self.readout, self.dipole, self.mean, self.std, self.scale = readout, False, None, None, None
        self.node_class = node_class
        self.max_num_neighbors = 32
        # ase.data.atomic_masses placeholder: only ever stored (schnet.py:47-48), real values come
        # from a loaded checkpoint
        self.register_buffer("atomic_mass", torch.zeros(119, dtype=torch.float64))
        self.embedding = nn.Embedding(node_class, hidden_channels)
        self.distance_expansion = GaussianSmearing(0.0, cutoff, num_gaussians)
        self.interactions = nn.ModuleList(
            [InteractionBlock(hidden_channels, num_gaussians, num_filters, cutoff) for _ in range(num_interactions)])
        self.lin1 = _nn.Linear(hidden_channels, hidden_channels)
        self.act = _nn.ShiftedSoftplus()
        self.lin2 = _nn.Linear(hidden_channels, hidden_channels)
        self.register_buffer("initial_atomref", None)
        self.atomref = None
        nn.init.xavier_uniform_(self.lin1.weight)
        self.lin1.bias.data.fill_(0)
        nn.init.xavier_uniform_(self.lin2.weight)
        self.lin2.bias.data.fill_(0)
        self.use_fused = True      # False: decomposed path (rbf kernel + library GEMMs + aggregate kernels)
        self.use_pairs = True      # False: per-edge fused kernels even where the pair form applies (cross-check)

    def _find_plan(self, z, batch):
        pl = _nn.lookup_plan(batch) if batch is not None else None
        if pl is None:
            b = torch.zeros(z.size(0), dtype=torch.long, device=z.device) if batch is None else batch
            d = types.SimpleNamespace(x=z, batch=b, num_graphs=int(b.max()) + 1 if b.numel() else 0,
                                      edge_index=None, edge_attr=None)
            pl = _plan.build_plan(d, max_nbr=self.max_num_neighbors, with_ext=False)
        return pl

    def forward(self, z, pos, batch=None, return_latent=False):
        assert z.dim() == 1 and z.dtype == torch.long
        pl = self._find_plan(z, batch)
        if pos.requires_grad and torch.is_grad_enabled():
            return self._forward_force_path(z, pos, pl, return_latent)
        ptr, nodes = _plan.z_lists(pl, self.node_class)
        h = hip.embedding_sum(self.embedding.weight, pl.z_codes, ptr, nodes)

        de = self.distance_expansion
        fusable = self.use_fused and self.num_filters == 128 and self.num_gaussians <= 64
        grad = torch.is_grad_enabled()
        # molecules of at most 33 atoms: the 32-neighbour cap cannot bind, the radius graph is symmetric and CFConv runs on
        # unordered pairs (half the filter-network work, no radius CSR at all); larger molecules keep the per-edge kernels
        pairwise = (fusable and hip.CFCONV_PAIR and self.use_pairs and self.num_gaussians <= 52
                    and pl.N_max <= self.max_num_neighbors + 1)
        Wfs = bwd_batch = None
        if pairwise:
            pp = hip.pair_plan(pos, pl, self.cutoff)
            if grad and hip.CFCONV_BWD_GROUP > 1:
                bwd_batch = hip.CfBwdBatch(pp, de.offset, de.coeff, self.cutoff, len(self.interactions))
            if hip.CFCONV_FILTER_MULTI and 0 < len(self.interactions) <= 8:
                # W = mlp(rbf(d)) * C(d) of every block (schnet.py:141-145) needs the distances only: one launch for all blocks,
                # in front of the layer chain (the chain is then lin1 -> aggregate -> lin2 -> lin per block)
                Wfs = hip.cfconv_pair_filters(pp, [(b_.mlp[0].weight, b_.mlp[0].bias, b_.mlp[2].weight, b_.mlp[2].bias)
                                                   for b_ in self.interactions], de.offset, de.coeff, self.cutoff)
        else:
            rplan, dist = hip.radius_plan(pos, pl.batch_i32, pl.mol_ptr, self.cutoff, pl.E_r_cap, self.max_num_neighbors,
                                           n_max=getattr(pl, "N_max", None))
        if not fusable:
            rbf, C = hip.rbf_cutoff(dist, rplan.E_dev, de.offset, de.coeff, self.cutoff)

        Hd, Fl = self.hidden_channels, self.num_filters
        nb = len(self.interactions)
        x1 = None
        for bi, blk in enumerate(self.interactions):
            if x1 is None:
                # h feeds the block and the residual: the fork lets the block's input-gradient GEMM accumulate the
                # residual's gradient (no separate add in the backward)
                h_res, x1 = _nn.linear_fork(h, blk.conv.lin1.weight)
            if pairwise and grad:
                agg = hip.cfconv_pair(x1, blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight, blk.mlp[2].bias, pp,
                                      de.offset, de.coeff, self.cutoff, Wf=None if Wfs is None else Wfs[bi], bwd_batch=bwd_batch)
            elif pairwise:
                agg, _ = hip.cfconv_pair_forward(x1, pp, blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight,
                                                 blk.mlp[2].bias, de.offset, de.coeff, self.cutoff,
                                                 Wf=None if Wfs is None else Wfs[bi])
            elif fusable and grad:
                agg = hip.cfconv_fused(x1, blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight, blk.mlp[2].bias,
                                       dist, rplan, de.offset, de.coeff, self.cutoff)
            elif fusable:
                agg = hip.cfconv_fused_forward(x1, dist, rplan, blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight,
                                               blk.mlp[2].bias, de.offset, de.coeff, self.cutoff)
            else:
                Wf = blk.mlp(rbf)
                agg = hip.cfconv_aggregate(x1, Wf, C, rplan)
            x1 = None
            if FUSE_TAIL and hip.rs_forward_ok(h.size(0), Hd, Fl, blk.conv.lin2.weight) \
                    and Hd % 4 == 0 and 0 < h.size(0) <= hip.RS_MAX_ROWS:
                # lin2 -> ssp -> lin -> + residual: two products with the pointwise stages in their epilogues
                h = hip.schnet_tail(agg, h_res, blk.conv.lin2, blk.lin)
            else:
                x = blk.conv.lin2(agg)
                x = blk.lin(hip.shifted_softplus(x))
                h = h_res + x

        if FUSE_TAIL and Hd % 4 == 0 and h.size(0) > 0:
            h = hip.mlp_fused(h, [(self.lin1.weight, self.lin1.bias), (self.lin2.weight, self.lin2.bias)], "ssp")
        else:
            h = self.lin2(hip.shifted_softplus(self.lin1(h)))
        out = hip.segment_reduce(h, pl.mol_ptr, pl.batch_i32, mean=(self.readout == "mean"))
        if return_latent:
            return out, h
        return out

    def _forward_force_path(self, z, pos, pl, return_latent):
        """Energy path that is differentiable w.r.t. positions, twice (finetune_MD17.py:47-78: forces by
        autograd.grad(create_graph=True), then backward through them).  Every operator between the positions and the
        energy -- coordinate differences, distances, smearing, cutoff, filter MLP, the dense layers, the message passing
        and the readout -- is a kernel-backed member of the closed operator set of moleculesde_amd.dd, so the first AND
        the second differentiation only launch library kernels.  The radius CSR comes from the radius kernels (indices
        carry no gradient) and the atom embedding, which does not depend on the positions, from the embedding kernel."""
        from .. import dd
        rplan, _ = hip.radius_plan(pos, pl.batch_i32, pl.mol_ptr, self.cutoff, pl.E_r_cap, self.max_num_neighbors,
                                   n_max=getattr(pl, "N_max", None))
        dist = dd.row_norm(dd.edge_diff(pos, rplan), rplan)
        de = self.distance_expansion
        rbf = dd.rbf(dist, rplan.src, de.offset, de.coeff)
        C = dd.cosine_cutoff(dist, self.cutoff, rplan.src)
        ptr, nodes = _plan.z_lists(pl, self.node_class)
        h = hip.embedding_sum(self.embedding.weight, pl.z_codes, ptr, nodes)
        # the smeared distances and the cutoff feed every block: one node each collects the blocks' gradients (dd.fanout)
        nb = len(self.interactions)
        rbfs, Cs = dd.fanout(rbf, nb), dd.fanout(C, nb)
        for bi, blk in enumerate(self.interactions):
            m0, m2 = blk.mlp[0], blk.mlp[2]
            rbf, C = rbfs[bi], Cs[bi]
            Wf = dd.mul_rows(dd.linear(dd.ssp(dd.linear(rbf, m0.weight, m0.bias)), m2.weight, m2.bias), C)
            x1 = dd.mm_nt(h, blk.conv.lin1.weight)
            agg = dd.edge_aggregate(x1, Wf, rplan)
            x = dd.linear(agg, blk.conv.lin2.weight, blk.conv.lin2.bias)
            h = dd.add(h, dd.linear(dd.ssp(x), blk.lin.weight, blk.lin.bias))
        h = dd.linear(dd.ssp(dd.linear(h, self.lin1.weight, self.lin1.bias)), self.lin2.weight, self.lin2.bias)
        out = dd.seg_reduce(h, pl.mol_ptr, pl.batch_i32, self.readout == "mean")
        if return_latent:
            return out, h
        return out

    def __repr__(self):
        return (f"{self.__class__.__name__}(hidden_channels={self.hidden_channels}, num_filters={self.num_filters}, "
                f"num_interactions={self.num_interactions}, num_gaussians={self.num_gaussians}, cutoff={self.cutoff})")
