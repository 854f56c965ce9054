"""PaiNN 3D encoder on HIP kernels.  Mirrors Geom3D/models/painn.py:118-269 (+ painn_utils.py): same constructor,
`forward(x, positions, radius_edge_index, batch, return_latent=False)` and state_dict keys (`embedding.weight`,
`filter_net.*`, `interactions.i.interatomic_context_net.{0,1}.*`, `mixing.i.intraatomic_context_net.{0,1}.*`,
`mixing.i.mu_channel_mix.weight`, buffers `cutoff_fn.cutoff`, `radial_basis.widths|offsets`).

Every arithmetic step between the positions and the output is an operator of the closed, twice-differentiable set of
moleculesde_amd.dd (kernels of csrc/dd.hip, csrc/gemm_ex.hip, the split-M weight-gradient kernel and the CSR edge
aggregation kernels), so PaiNN also serves the MD17 force objective (finetune_MD17.py:47-78).  Layout choices:
  * the vector channel mu [N, 3, F] (painn.py:241) is kept as three [N, F] tensors, one per Cartesian component; per-edge
    unit vectors as three contiguous [E] vectors: no reshapes, every kernel sees contiguous rows;
  * a Dense whose output is `torch.split` three ways (painn.py:57,109) runs as three products on ROW blocks of its weight
    (contiguous views), so no strided column slices exist;
  * the neighbour sums (scatter_add over idx_i, painn.py:58,60) are fixed-order segmented sums over a by-target CSR of
    the given radius graph:  dq_i = sum_j xq_j Wq_ij,  dmu_i,c = sum_j xR_j (WR_ij dir_ij,c) + sum_j (xm_j mu_j,c) Wm_ij.
What stays on torch: slicing parameters into row blocks and the one concatenation of painn.py:106 (views / one copy,
differentiable at any order); building the CSR of a caller-supplied edge list (sort, once per batch).
"""
import math
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import dd, hip, plan as _plan
from . import nn as _nn


class Dense(nn.Linear):
    """painn_utils.py:9-35 (Xavier weight, zero bias, optional activation); the product runs in PaiNN.forward."""

    def __init__(self, in_features, out_features, bias=True, activation=None):
        super().__init__(in_features, out_features, bias)
        self.activation = activation if activation is not None else nn.Identity()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)


class GaussianRBF(nn.Module):
    def __init__(self, n_rbf, cutoff, start=0.0):
        super().__init__()
        self.n_rbf = n_rbf
        offset = torch.linspace(start, cutoff, n_rbf)
        self.register_buffer("widths", torch.abs(offset[1] - offset[0]) * torch.ones_like(offset))
        self.register_buffer("offsets", offset)


class CosineCutoff(nn.Module):
    def __init__(self, cutoff):
        super().__init__()
        self.register_buffer("cutoff", torch.FloatTensor([cutoff]))


class PaiNNInteraction(nn.Module):
    def __init__(self, n_atom_basis, activation):
        super().__init__()
        self.n_atom_basis = n_atom_basis
        self.interatomic_context_net = nn.Sequential(Dense(n_atom_basis, n_atom_basis, activation=activation),
                                                     Dense(n_atom_basis, 3 * n_atom_basis, activation=None))


class PaiNNMixing(nn.Module):
    def __init__(self, n_atom_basis, activation, epsilon=1e-8):
        super().__init__()
        self.n_atom_basis = n_atom_basis
        self.intraatomic_context_net = nn.Sequential(Dense(2 * n_atom_basis, n_atom_basis, activation=activation),
                                                     Dense(n_atom_basis, 3 * n_atom_basis, activation=None))
        self.mu_channel_mix = Dense(n_atom_basis, 2 * n_atom_basis, activation=None, bias=False)
        self.epsilon = epsilon


def _blocks(dense, n, F_):
    """Row blocks of a Dense whose output the reference splits n ways: [(W_k [F, in], b_k [F])]"""
    return [(dense.weight[k * F_:(k + 1) * F_], None if dense.bias is None else dense.bias[k * F_:(k + 1) * F_])
            for k in range(n)]


class PaiNN(nn.Module):
    def __init__(self, n_atom_basis, n_interactions, n_rbf, cutoff, n_out, readout, n_out_hidden=None, n_out_layers=2,
                 activation=F.silu, max_z=100, shared_interactions=False, shared_filters=False, epsilon=1e-8):
        super().__init__()
        if activation is not F.silu:
            raise NotImplementedError("PaiNN is used with SiLU everywhere in MoleculeSDE (painn.py:135)")
        self.n_atom_basis, self.n_interactions, self.n_out = n_atom_basis, n_interactions, n_out
        self.n_out_hidden, self.n_out_layers, self.activation = n_out_hidden, n_out_layers, activation
        self.cutoff = cutoff
        self.cutoff_fn = CosineCutoff(cutoff)
        self.radial_basis = GaussianRBF(n_rbf=n_rbf, cutoff=cutoff)
        self.readout = readout
        self.max_z = max_z
        self.embedding = nn.Embedding(max_z, n_atom_basis, padding_idx=0)
        self.share_filters = shared_filters
        self.filter_net = Dense(n_rbf, 3 * n_atom_basis if shared_filters else n_interactions * n_atom_basis * 3)
        mk_i = lambda: PaiNNInteraction(n_atom_basis, activation)
        mk_m = lambda: PaiNNMixing(n_atom_basis, activation, epsilon)
        self.interactions = nn.ModuleList([mk_i()] * n_interactions if shared_interactions else
                                          [mk_i() for _ in range(n_interactions)])
        self.mixing = nn.ModuleList([mk_m()] * n_interactions if shared_interactions else
                                    [mk_m() for _ in range(n_interactions)])
        # nn.Embedding(padding_idx=0): row 0 starts at zero and never receives a gradient
        self.embedding.weight.register_hook(self._mask_padding_grad)

    @staticmethod
    def _mask_padding_grad(g):
        g = g.clone()
        g[0].zero_()
        return g

    def create_output_layers(self):
        """painn.py:208-216 / painn_utils.py:38-70"""
        n_in, n_out, n_layers = self.n_atom_basis, self.n_out, self.n_out_layers
        if self.n_out_hidden is None:
            c, neurons = n_in, []
            for _ in range(n_layers):
                neurons.append(c)
                c = max(n_out, c // 2)
            neurons.append(n_out)
        else:
            hidden = [self.n_out_hidden] * (n_layers - 1) if isinstance(self.n_out_hidden, int) else list(self.n_out_hidden)
            neurons = [n_in] + hidden + [n_out]
        layers = [_OutDense(neurons[i], neurons[i + 1], activation=self.activation) for i in range(n_layers - 1)]
        layers.append(_OutDense(neurons[-2], neurons[-1], activation=None))
        return nn.Sequential(*layers)

    # ------------------------------------------------------------------------------------------------------------
    def _edge_plan(self, radius_edge_index, n_atoms):
        """by-target CSR of the given radius graph: centre i = row 0, neighbour j = row 1 (painn.py:235)"""
        key = (radius_edge_index.data_ptr(), tuple(radius_edge_index.shape), n_atoms)
        cached = getattr(self, "_plan_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        pl = hip.build_csr(torch.stack([radius_edge_index[1], radius_edge_index[0]]), n_atoms)   # messages j -> i
        self._plan_cache = (key, pl)
        return pl

    def forward(self, x, positions, radius_edge_index, batch, return_latent=False):
        z = x[:, 0] if x.dim() == 2 else x
        if not positions.is_cuda:
            raise RuntimeError("moleculesde_amd.geom3d.PaiNN runs on the HIP device only (no CPU fallback)")
        N, Fd = z.size(0), self.n_atom_basis
        ep = self._edge_plan(radius_edge_index, N)
        bp = _nn.lookup_plan(batch)
        if bp is None or not hasattr(bp, "z_codes"):
            d = types.SimpleNamespace(x=z, batch=batch, num_graphs=int(batch.max()) + 1 if batch.numel() else 0,
                                      edge_index=None, edge_attr=None)
            bp = _plan.build_plan(d, with_ext=False)

        # geometry (painn.py:237-244): r_ij = pos_i - pos_j; dd.edge_diff gives pos[src] - pos[dst] = pos_j - pos_i
        r = dd.scale(dd.edge_diff(positions, ep), -1.0)                       # [E, 3]
        d_ij = dd.row_norm(r, ep)                                            # [E]
        dirs = dd.components(dd.mul_rows(r, dd.recip(d_ij)))                 # 3 x [E]
        rb = self.radial_basis
        coeff = -0.5 / float(rb.widths[0]) ** 2
        phi = dd.rbf(d_ij, None, rb.offsets, coeff)                          # [E, n_rbf]
        fcut = dd.cosine_cutoff(d_ij, float(self.cutoff), None)              # [E], zero beyond the cutoff

        ptr, nodes = _plan.z_lists(bp, self.max_z)
        q = hip.embedding_sum(self.embedding.weight, bp.z_codes, ptr, nodes)  # [N, F]
        mu = None                                                            # zeros (painn.py:248)
        nblk = 1 if self.share_filters else self.n_interactions
        fblocks = _blocks(self.filter_net, 3 * nblk, Fd)
        for i, (inter, mix) in enumerate(zip(self.interactions, self.mixing)):
            # ---- filters of this block (painn.py:246-251): Dense(phi) * fcut, one product per split part
            fb = fblocks[0:3] if self.share_filters else fblocks[3 * i:3 * i + 3]
            Wq, WR, Wm = (dd.mul_rows(dd.linear(phi, w, b), fcut) for w, b in fb)
            # ---- inter-atomic (painn.py:52-65)
            net = inter.interatomic_context_net
            h = dd.silu(dd.linear(q, net[0].weight, net[0].bias))
            xq, xR, xm = (dd.linear(h, w, b) for w, b in _blocks(net[1], 3, Fd))
            dq = dd.edge_aggregate(xq, Wq, ep)
            new_mu = []
            for c in range(3):
                dmu = dd.edge_aggregate(xR, dd.mul_rows(WR, dirs[c]), ep)
                if mu is not None:
                    dmu = dd.add(dmu, dd.edge_aggregate(dd.mul(xm, mu[c]), Wm, ep))
                    dmu = dd.add(mu[c], dmu)
                new_mu.append(dmu)
            q, mu = dd.add(q, dq), new_mu
            # ---- intra-atomic mixing (painn.py:100-116)
            (wv, _), (ww, _) = _blocks(mix.mu_channel_mix, 2, Fd)
            V = [dd.mm_nt(mu[c], wv) for c in range(3)]
            Wc = [dd.mm_nt(mu[c], ww) for c in range(3)]
            s = dd.add(dd.add(dd.mul(V[0], V[0]), dd.mul(V[1], V[1])), dd.mul(V[2], V[2]))
            ctx = torch.cat([q, dd.sqrt_eps(s, mix.epsilon)], dim=-1)
            net = mix.intraatomic_context_net
            h = dd.silu(dd.linear(ctx, net[0].weight, net[0].bias))
            dq_i, dmu_i, dqmu_i = (dd.linear(h, w, b) for w, b in _blocks(net[1], 3, Fd))
            dot = dd.add(dd.add(dd.mul(V[0], Wc[0]), dd.mul(V[1], Wc[1])), dd.mul(V[2], Wc[2]))
            q = dd.add(dd.add(q, dq_i), dd.mul(dqmu_i, dot))
            mu = [dd.add(mu[c], dd.mul(dmu_i, Wc[c])) for c in range(3)]

        if self.readout not in ("mean", "sum", "add"):
            raise NotImplementedError(f"readout={self.readout!r}")
        h = dd.seg_reduce(q, bp.mol_ptr, bp.batch_i32, self.readout == "mean")
        if return_latent:
            return h, q
        return h


class _OutDense(Dense):
    """Dense of the caller-side output head (create_output_layers): a plain module with its own forward."""

    def forward(self, input):
        return self.activation(_nn.linear(input, self.weight, self.bias))      # HIP only: _nn.linear refuses host tensors
