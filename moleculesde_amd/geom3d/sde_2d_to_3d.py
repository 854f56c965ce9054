"""SDEModel2Dto3D_02 + EquivariantScoreNetwork on HIP kernels.

API, constructor arguments, `forward(node_2D_repr, data, anneal_power) -> {"position": loss}`,
`get_score(...)`, `.sde_pos` and state_dict keys mirror SDE_model_2D_to_3D.py:252-445 and
equivariant_scorenetwork.py:13-169.

What runs where:
  * per-edge SE(3) frame, distance, pseudo-angle and all Gaussian-Fourier features: one kernel
    (hip.edge_geometry); no gradient flows to coordinates;
  * `edge_2D_emb`'s Linear(cat(h_row, h_col)) is factored into two node-level GEMMs plus a
    gather-add kernel (12.3 -> 1.3 GFLOP at bs 256);
  * TransformerConv = q/k/v/skip/edge Linear (library GEMMs) + hip.edge_attention (scores,
    per-target softmax, dropout, weighted sum in one kernel);
  * basis mixing + mean scatter: hip.frame_mix_mean.
Edge tensors live in the plan's canonical (by-target) order; nothing per-edge is returned to the
caller, so the order is internal.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import dd as _dd, escore as _escore, hip, plan as _plan, wcache
from . import nn as _nn
from .sde import VESDE, VPSDE

EPSILON = 1e-6
import os as _os
FUSE_GAT_TAIL = True     # csrc/gat_tail.hip for hidden size 32 (False: the kernel-per-stage path, used as a cross-check)
NOISE_IN_KERNEL = True     # DeviceNoise: draws made by the VE perturbation kernel
FUSE_FRAME = True          # hip._FrameMLP (False: coff_mlp twice + cat + project)
STATIC_FEATURE_CACHE = True   # inference: coordinate-independent inputs of the score network computed once per 2D representation
FUSE_HEAD_MIX = True       # hip._MlpHeadMix (False: hip.mlp_fused + hip.frame_mix_mean)
FUSE_EDGE_EMB = True       # hip._PairBnReluLinear (False: gather-add, BatchNorm, Linear as separate ops)
MOL_KERNEL = True       # EquivariantScoreNetwork without autograd (get_score, sampling, evaluation) as ONE launch, one workgroup per molecule (False: operator by operator, the cross-check)
MOL_KERNEL_SCORE = True  # get_score on msde_escore_mol_score: edge features, lin_edge x 4 and the basis MLPs' edge halves in ONE wide launch, the per-molecule chain behind it (cached 2D features)
MOL_KERNEL_TRAIN = False    # ... and under autograd (forward + one-launch backward, moleculesde_amd/escore.py).  Off by default: beside
                            # the second stream of the pretrain step the 256 single-wave-per-SIMD workgroups hold every CU for ~100 + ~340 us
                            # and the step is 2.69 ms against 2.58 ms operator by operator (alternating A/B on one box);
                            # `--score_kernel mol` of pretrain.py / the parity tests switch it on
FANOUT_EDGE_ATTR = True    # the score network's edge features through ONE fan-out node (dd.fanout): their 3 consumers' gradients summed by one launch
FUSE_PAIR_LINEAR = True    # hip._PairLinear (False: re-laid-out weight per step)


class GaussianFourierProjection(nn.Module):
    """Frozen random features W (in the state dict, excluded from the optimiser; App. B.10)."""

    def __init__(self, embedding_size, scale=1.0):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embedding_size) * scale, requires_grad=False)


class TransformerConv(nn.Module):
    """PyG TransformerConv(in, out, heads, dropout, edge_dim) parameters (App. A.4 / C.2)."""

    def __init__(self, in_channels, out_channels, heads, dropout, edge_dim):
        super().__init__()
        self.heads, self.out_channels, self.dropout = heads, out_channels, dropout
        self.lin_key = _nn.Linear(in_channels, heads * out_channels)
        self.lin_query = _nn.Linear(in_channels, heads * out_channels)
        self.lin_value = _nn.Linear(in_channels, heads * out_channels)
        self.lin_edge = _nn.Linear(edge_dim, heads * out_channels, bias=False)
        self.lin_skip = _nn.Linear(in_channels, heads * out_channels, bias=True)

    def fusion_sets(self):
        return [[self.lin_query.weight, self.lin_key.weight, self.lin_value.weight, self.lin_skip.weight],
                [self.lin_query.bias, self.lin_key.bias, self.lin_value.bias, self.lin_skip.bias]]

    def forward(self, x, edge_attr, plan, seed, seed_dev=None, ee_all=None, col=0, shared=None):
        # one projection GEMM for query | key | value | skip (they share the input x)
        Ws, bs = self.fusion_sets()
        W, b = hip.cat_params(Ws), hip.cat_params(bs)     # free views once FlatAdam has laid them out back to back
        x_res, qkvs = _nn.linear_fork(x, W, b)      # x also feeds the caller's residual
        p = self.dropout if self.training else 0.0
        if ee_all is not None:           # lin_edge of all layers evaluated as one GEMM by the caller
            return x_res, hip.edge_attention_fused(qkvs, ee_all, plan, self.heads, p, seed, seed_dev, col, shared)
        ee = self.lin_edge(edge_attr)
        return x_res, hip.edge_attention_fused(qkvs, ee, plan, self.heads, p, seed, seed_dev)


class GATLayer(nn.Module):
    def __init__(self, n_head, hidden_dim, dropout=0.2):
        super().__init__()
        assert hidden_dim % n_head == 0
        self.MHA = TransformerConv(hidden_dim, hidden_dim // n_head, n_head, dropout, hidden_dim)
        self.FFN = nn.Sequential(_nn.Linear(hidden_dim, hidden_dim), nn.SiLU(), nn.Dropout(dropout),
                                 _nn.Linear(hidden_dim, hidden_dim))
        self.norm1 = nn.LayerNorm(hidden_dim)
        self.norm2 = nn.LayerNorm(hidden_dim)

    def forward(self, plan, node_attr, edge_attr, seed, seed_dev=None, ee_all=None, col=0, shared=None, silu_out=False):
        """silu_out: apply the SiLU that follows every convolution but the last of a block
        (equivariant_scorenetwork.py:142) inside the layer, so it can ride in the fused kernel."""
        node_attr, x = self.MHA(node_attr, edge_attr, plan, seed, seed_dev, ee_all, col, shared)
        p = self.FFN[2].p if self.training else 0.0
        if x.size(-1) == 32 and FUSE_GAT_TAIL:
            # LayerNorm + residual, feed-forward, LayerNorm + residual (+ SiLU): one kernel each way
            return hip.gat_tail(x, node_attr, self.norm1, self.FFN[0], self.FFN[3], self.norm2, p, seed ^ 0x46464E,
                                seed_dev, silu_out)
        if x.size(-1) % 4 == 0:
            node_attr = hip.res_layernorm(x, node_attr, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            # FFN = Linear -> SiLU -> Dropout -> Linear with the two pointwise stages in one kernel
            node_attr, f0 = self.FFN[0].fork(node_attr)
            x = self.FFN[3](hip.silu_dropout(f0, p, seed ^ 0x46464E, seed_dev))
            out = hip.res_layernorm(x, node_attr, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            return hip.silu_dropout(out) if silu_out else out
        node_attr = node_attr + self.norm1(x)
        x = self.FFN(node_attr)
        out = node_attr + self.norm2(x)
        return F.silu(out) if silu_out else out


class _EquiLayer(nn.Module):
    """Holds the `eps` buffer of the reference's EquiLayer (state-dict key only)."""

    def __init__(self):
        super().__init__()
        self.register_buffer("eps", torch.Tensor([0.0]))


class EquivariantScoreNetwork(nn.Module):
    def __init__(self, hidden_dim, hidden_coff_dim=64, activation="silu", short_cut=False, concat_hidden=False):
        super().__init__()
        if short_cut or concat_hidden:
            raise NotImplementedError("short_cut / concat_hidden are never enabled on the MoleculeSDE path")
        self.hidden_dim, self.num_layers, self.num_convs = hidden_dim, 2, 2
        self.num_head, self.dropout, self.hidden_coff_dim = 8, 0.1, hidden_coff_dim
        self.gnn_layers = nn.ModuleList()
        self.equi_modules = nn.ModuleList()
        self.basis_mlp_modules = nn.ModuleList()
        for _ in range(self.num_layers):
            self.gnn_layers.append(nn.ModuleList(
                [GATLayer(self.num_head, hidden_dim, dropout=self.dropout) for _ in range(self.num_convs)]))
            self.equi_modules.append(_EquiLayer())
            self.basis_mlp_modules.append(nn.Sequential(
                _nn.Linear(2 * hidden_dim, hidden_coff_dim), nn.SiLU(), _nn.Linear(hidden_coff_dim, 3)))
        self._seed_base = 0x5DE2D3D
        self._calls = 0
        self.seed_dev = None   # device uint64 step counter (set by the trainer for hipGraph replay)
        self.mol_kernel_train = None   # None: the module default MOL_KERNEL_TRAIN; the trainer sets it from --score_kernel

    def fusion_sets(self):
        # lin_edge of every GAT layer consumes the same edge features: one stacked weight, one GEMM
        return [[gnn.MHA.lin_edge.weight for layers in self.gnn_layers for gnn in layers]]

    def forward(self, plan, node_attr, edge_attr, basis, pl=None):
        """pl: the batch plan (atom ranges of the molecules) -- given, and the shapes allowing, the whole network is ONE launch
        with one workgroup per molecule (moleculesde_amd/escore.py, csrc/escore_mol.hip); otherwise operator by operator."""
        if self.seed_dev is None:
            self._calls += 1     # eager: host-side call counter; graph mode: the device counter varies the mask
        train_mol = MOL_KERNEL_TRAIN if self.mol_kernel_train is None else self.mol_kernel_train
        if ((train_mol if torch.is_grad_enabled() else MOL_KERNEL) and node_attr.is_cuda and plan.E > 0
                and _escore.supported(self, pl)):
            seed0 = (self._seed_base + self._calls) * 16
            return {"node_feature": None,
                    "gradient": _escore.forward(self, plan, pl, node_attr, edge_attr, basis, seed0, self.seed_dev)}
        conv_input = node_attr
        gradient = None
        ee_all, shared, D = None, None, self.hidden_dim
        # edge_attr has 1 + len(gnn_layers) consumers (the stacked lin_edge product, every basis MLP's input): one fan-out node
        # whose backward sums their gradients -- two of them column blocks of wider buffers -- in ONE launch instead of one
        # autograd addition per extra consumer on the backward chain
        n_use = 1 + len(self.gnn_layers)
        ea = (_dd.fanout(edge_attr, n_use) if (FANOUT_EDGE_ATTR and n_use > 2 and edge_attr.is_cuda and torch.is_grad_enabled()
                                              and edge_attr.requires_grad) else (edge_attr,) * n_use)
        ee_all = _nn.linear(ea[0], hip.cat_params(self.fusion_sets()[0]))       # [E, layers*D]
        shared = {}
        layer_no = 0
        for module_idx, gnn_layers in enumerate(self.gnn_layers):
            for conv_idx, gnn in enumerate(gnn_layers):
                seed = (self._seed_base + self._calls) * 16 + module_idx * 4 + conv_idx
                hidden = gnn(plan, conv_input, ea[0], seed, self.seed_dev, ee_all, layer_no * D, shared,
                             silu_out=conv_idx < len(gnn_layers) - 1)
                layer_no += 1
                conv_input = hidden
            node_feature = hidden
            mlp = self.basis_mlp_modules[module_idx]
            edge_attr = ea[1 + module_idx]
            if _nn.FUSED_MLP and node_feature.size(1) % 4 == 0 and edge_attr.size(1) % 4 == 0:
                # cat([h_row + h_col, edge_attr]) written by the gather; Linear -> SiLU -> Linear on gemm_ex epilogues
                edge_feature = hip.pair_gather_cat(node_feature, edge_attr, plan)
                if FUSE_HEAD_MIX and plan.E > 0 and hip.mlp_head_mix_ok(edge_feature, mlp[0], mlp[2]):
                    # head + frame mix + mean + running sum in one kernel; the coefficients are never stored
                    gradient = hip.mlp_head_mix(edge_feature, mlp[0], mlp[2], basis, plan, gradient)
                    continue
                coff = hip.mlp_fused(edge_feature, [(mlp[0].weight, mlp[0].bias), (mlp[2].weight, mlp[2].bias)], "silu")
            else:
                pair = hip.pair_gather_add(node_feature, node_feature, plan)        # h_row + h_col
                edge_feature = torch.cat([pair, edge_attr], dim=-1)
                coff = mlp(edge_feature)                                           # [E, 3]
            gradient = hip.frame_mix_mean(coff, basis, plan, gradient)      # `gradient += ...` folded into the kernel
        return {"node_feature": node_feature, "gradient": gradient}


class SDEModel2Dto3D_02(nn.Module):
    def __init__(self, emb_dim, hidden_dim, beta_schedule, beta_min, beta_max, num_diffusion_timesteps,
                 SDE_type="VE", short_cut=False, concat_hidden=False, use_extend_graph=False):
        super().__init__()
        self.emb_dim, self.hidden_dim = emb_dim, hidden_dim
        self.SDE_type, self.use_extend_graph = SDE_type, use_extend_graph
        self.node_emb = _nn.MultiLayerPerceptron(emb_dim, [hidden_dim], activation="silu")
        bn = _nn.BatchNorm1d(emb_dim)
        bn.fuse_relu = True                      # edge_2D_emb[2] (ReLU) is fused into the BatchNorm kernel
        self.edge_2D_emb = nn.Sequential(_nn.Linear(emb_dim * 2, emb_dim), bn, nn.Identity(),
                                         _nn.Linear(emb_dim, hidden_dim))
        self.dist_gaussian_fourier = GaussianFourierProjection(hidden_dim, scale=1)
        self.input_mlp = _nn.MultiLayerPerceptron(2 * hidden_dim, [hidden_dim], activation="silu")
        self.coff_gaussian_fourier = GaussianFourierProjection(hidden_dim, scale=1)
        self.coff_mlp = _nn.Linear(4 * hidden_dim, hidden_dim)
        self.coff_mlp.shared = True            # applied to both frame features of an edge
        self.project = _nn.MultiLayerPerceptron(2 * hidden_dim + 2, [hidden_dim, hidden_dim], activation="silu")
        self.score_network = EquivariantScoreNetwork(hidden_dim, hidden_coff_dim=128, activation="silu",
                                                     short_cut=short_cut, concat_hidden=concat_hidden)
        if SDE_type in ("VE", "VE_test"):
            self.sde_pos = VESDE(sigma_min=beta_min, sigma_max=beta_max, N=num_diffusion_timesteps)
        elif SDE_type in ("VP", "VP_test"):
            self.sde_pos = VPSDE(beta_min=beta_min, beta_max=beta_max, N=num_diffusion_timesteps)
        else:
            raise NotImplementedError(f"SDE_type={SDE_type!r}")
        self.num_diffusion_timesteps = num_diffusion_timesteps
        self.noise = _nn.DeviceNoise()     # set to nn.CpuReplayNoise(seed) for replayable parity runs
        self.register_buffer("_zero_bias", torch.zeros(emb_dim), persistent=False)
        self.side_stream = None            # optional second HIP stream for the coordinate-only branch
        self._pending = None               # results of begin() waiting for forward()
        self._pending_event = None         # recorded by a caller that ran begin() on ANOTHER stream (see forward)

    def _plan(self, data):
        pl = _plan.get_plan(data)
        return pl, (pl.ext if self.use_extend_graph else pl.bond)

    def _geometry_branch(self, pos_perturbed, ep):
        """Everything that depends on coordinates only (frame, Fourier features, their MLPs)."""
        has_dist = hasattr(self, "input_mlp")
        H = self.hidden_dim
        if (FUSE_FRAME and H % 4 == 0 and pos_perturbed.is_cuda and len(self.project.layers) == 2
                and self.project.activation_name == "silu" and not self.project.dropout and ep.E > 0):
            # frame features stacked [feat_i; feat_j]: the shared coff_mlp runs once, project[0] reads its result and the
            # pseudo-angle in place (no cat), SiLU in the product's epilogue -- hip._FrameMLP
            feat_d, feat, X, basis = hip.edge_geometry_stacked(
                pos_perturbed, ep, (self.dist_gaussian_fourier if has_dist else self.coff_gaussian_fourier).W,
                self.coff_gaussian_fourier.W, H)
            edge_attr_3D_invariant = self.input_mlp(feat_d) if has_dist else None
            frame = hip.frame_mlp(feat, X, self.coff_mlp, self.project.layers[0], self.project.layers[1])
            return edge_attr_3D_invariant, frame, basis
        feat_d, feat_i, feat_j, angle, basis = hip.edge_geometry(
            pos_perturbed, ep, (self.dist_gaussian_fourier if has_dist else self.coff_gaussian_fourier).W,
            self.coff_gaussian_fourier.W)
        edge_attr_3D_invariant = self.input_mlp(feat_d) if has_dist else None
        embed_i = self.coff_mlp(feat_i)
        embed_j = self.coff_mlp(feat_j)
        edge_attr_3D_frame_invariant = self.project(torch.cat([angle, embed_i, embed_j], dim=-1))
        return edge_attr_3D_invariant, edge_attr_3D_frame_invariant, basis

    def _launch_geometry(self, pos_perturbed, ep):
        """Start the coordinate-only branch: on the side stream if the trainer gave us one (it is independent of
        the 2D representation, so it runs beside the GIN encoder / the 2D-embedding branch; autograd mirrors the
        overlap in the backward), else inline.  Returns (tensors, side stream used or None)."""
        side = self.side_stream
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                geo = self._geometry_branch(pos_perturbed, ep)
        else:
            geo = self._geometry_branch(pos_perturbed, ep)
        return geo, side

    def begin(self, data):
        """Optional early start (the trainer calls it BEFORE the 2D encoder runs): draws this step's noise --
        same draws, same order as forward() -- perturbs the coordinates and launches the coordinate-only branch,
        which then overlaps the GIN forward.  forward(node_2D_repr, data, ...) picks the results up."""
        pos = data.positions
        pl, ep = self._plan(data)
        B = data.num_graphs
        T = self.num_diffusion_timesteps
        if (NOISE_IN_KERNEL and self.SDE_type == "VE" and pos.is_cuda and getattr(self.noise, "draws_in_kernel", None)
                and self.noise.draws_in_kernel()):
            # position noise and time steps drawn inside the perturbation kernel: one launch instead of three
            pos_noise, pos_perturbed, std_pos = hip.ve_perturb_rng(
                pos, pl.batch_i32, B, T, EPSILON, self.sde_pos.sigma_min, self.sde_pos.sigma_max, self.noise.next_seed(),
                self.noise.seed_dev)
            geo, side = self._launch_geometry(pos_perturbed, ep)
            self._pending = (data, pos_noise, std_pos, pos_perturbed, geo, side)
            return
        pos_noise = self.noise.randn_like(pos)
        draws = self.noise.randint(T, (B // 2 + 1,), pos.device)
        if self.SDE_type == "VE" and draws.dtype == torch.int64:
            pos_perturbed, std_pos = hip.ve_perturb(pos, pos_noise, draws, pl.batch_i32, B, T, EPSILON,
                                                    self.sde_pos.sigma_min, self.sde_pos.sigma_max)
        else:
            time_step = torch.cat([draws, T - draws - 1], dim=0)[:B]
            if self.SDE_type in ("VE", "VP"):
                time_step = time_step / T * (1 - EPSILON) + EPSILON
            t_pos = time_step.index_select(0, data.batch)
            mean_pos, std_pos = self.sde_pos.marGINal_prob(pos.detach(), t_pos)
            pos_perturbed = mean_pos + std_pos[:, None] * pos_noise
        geo, side = self._launch_geometry(pos_perturbed, ep)
        self._pending = (data, pos_noise, std_pos, pos_perturbed, geo, side)

    def _static_features(self, node_2D_repr, ep):
        """(edge_2D_emb of the node pairs, node_emb(node_2D_repr)): the part of the model's input that does not depend on the
        coordinates.  A sampler calls get_score thousands of times with the SAME 2D representation (2 calls per predictor-
        corrector iteration, pretrain_MoleculeSDE_inference_2D_to_3D_VE_VP.py:92-138): in eval mode without autograd the pair
        is computed once per (representation, graph, parameter epoch) and reused -- ~12 launches less per score call, and none
        of them inside a captured iteration."""
        key = (node_2D_repr.data_ptr(), node_2D_repr._version, tuple(node_2D_repr.shape), id(ep), ep.E, ep.N, ep.src.data_ptr(),
               wcache.weight_epoch(),
               tuple(t._version for m in (self.edge_2D_emb, self.node_emb) for t in list(m.parameters()) + list(m.buffers())))
        hit = getattr(self, "_static_cache", None)
        # the entry PINS the representation and the plan it was computed from and is matched by identity: addresses and
        # id()s alone are reused by the allocator / CPython once the previous molecule's objects die
        if hit is not None and hit[0] == key and hit[3] is node_2D_repr and hit[4] is ep:
            return hit[1], hit[2]
        if torch.cuda.is_current_stream_capturing():
            return None            # (computed inside the capture like everything else; cached by the next eager call)
        edge_attr_2D = self._edge_2D(node_2D_repr, ep)
        node_attr = self.node_emb(node_2D_repr)
        self._static_cache = (key, edge_attr_2D, node_attr, node_2D_repr, ep)
        return edge_attr_2D, node_attr

    def _edge_2D(self, node_2D_repr, ep):
        D = self.emb_dim
        lin0 = self.edge_2D_emb[0]
        if FUSE_PAIR_LINEAR and torch.is_grad_enabled() and hip.pair_linear_ok(node_2D_repr, lin0):
            AB = hip.pair_linear(node_2D_repr, lin0)          # stacked weight copy cached per optimiser step
        else:
            Wst = lin0.weight.view(D, 2, D).transpose(0, 1).reshape(2 * D, D)
            bst = torch.cat([self._zero_bias, lin0.bias])          # the row half carries no bias
            AB = _nn.linear(node_2D_repr, Wst, bst)
        bn, lin3 = self.edge_2D_emb[1], self.edge_2D_emb[3]
        if (FUSE_EDGE_EMB and torch.is_grad_enabled() and _nn.bn_fusable(bn) and bn.fuse_relu and ep.E > 0
                and hip.pair_bn_relu_linear_ok(AB, bn, lin3)):
            # gather-add + BatchNorm statistics in one pass, BatchNorm + ReLU inside the second Linear (hip._PairBnReluLinear)
            out = hip.pair_bn_relu_linear(AB, ep, bn, lin3)
            _nn.count_batch(bn)
            return out
        pre = hip.pair_gather_add_cols(AB, ep)
        return lin3(self.edge_2D_emb[2](bn(pre)))

    def _edge_and_node_features(self, node_2D_repr, pos_perturbed, ep, started=None):
        D = self.emb_dim
        # (a pair: the caller's fan-out aliases of the representation for the edge and the node branch, pretrain.Trainer.losses)
        repr_edge, repr_node = node_2D_repr if isinstance(node_2D_repr, tuple) else (node_2D_repr, node_2D_repr)
        geo, side = started if started is not None else self._launch_geometry(pos_perturbed, ep)
        edge_attr_3D_invariant, edge_attr_3D_frame_invariant, basis = geo
        static = None
        if not self.training and not torch.is_grad_enabled() and STATIC_FEATURE_CACHE:
            static = self._static_features(repr_node, ep)
        # edge_2D_emb[0](cat(h[row], h[col])) == h[row] W[:, :D]^T + h[col] W[:, D:]^T + b
        # as ONE node-level GEMM with the two weight halves stacked ([W_row; W_col], one re-layout copy per step): _edge_2D
        edge_attr_2D = static[0] if static is not None else self._edge_2D(repr_edge, ep)
        if side is not None:
            main = torch.cuda.current_stream()
            main.wait_stream(side)
            for t in (edge_attr_3D_invariant, edge_attr_3D_frame_invariant, basis):
                if t is not None:
                    t.record_stream(main)
        if edge_attr_3D_invariant is None:         # SDEModel2Dto3D_01: no distance branch (:181)
            edge_attr = edge_attr_2D + edge_attr_3D_frame_invariant
        else:
            edge_attr = hip.mul_add(edge_attr_3D_invariant, edge_attr_2D, edge_attr_3D_frame_invariant)
        node_attr = static[1] if static is not None else self.node_emb(repr_node)
        return node_attr, edge_attr, basis

    def forward(self, node_2D_repr, data, anneal_power):
        pl, ep = self._plan(data)
        if self._pending is None or self._pending[0] is not data:
            self.begin(data)
        _, pos_noise, std_pos, pos_perturbed, geo, side = self._pending
        self._pending = None
        if self._pending_event is not None:
            # begin() ran on another stream (the trainer puts the whole coordinate-only branch -- noise, perturbation,
            # frame / Fourier features and their MLPs -- at the head of its second stream): join by EVENT, so that the
            # work queued on that stream afterwards (SchNet) is not waited for
            cur = torch.cuda.current_stream()
            cur.wait_event(self._pending_event)
            self._pending_event = None
            for t in (pos_noise, std_pos, pos_perturbed) + tuple(geo):
                if isinstance(t, torch.Tensor):
                    t.record_stream(cur)

        node_attr, edge_attr, basis = self._edge_and_node_features(node_2D_repr, pos_perturbed, ep, (geo, side))
        scores = self.score_network(ep, node_attr, edge_attr, basis, pl)["gradient"]
        # sum_k (score - noise)^2 [* std^anneal_power] -> scatter_mean over molecules -> mean: one kernel pair
        return {"position": hip.ve_position_loss(scores, pos_noise, std_pos, anneal_power, pl.mol_ptr, pl.batch_i32)}

    def _score_fused(self, node_2D_repr, pos_perturbed, pl, ep):
        """get_score's network evaluation as TWO launches (msde_escore_mol_score): frame / Fourier features / input_mlp /
        coff_mlp / project, lin_edge of every layer and the edge halves of the basis MLPs from the coordinates and the CACHED 2D
        edge features in a wide launch; the per-molecule chain behind it.
        None when the shapes do not allow it (the caller then runs operator by operator + msde_escore_mol_fwd)."""
        if not (MOL_KERNEL and MOL_KERNEL_SCORE and STATIC_FEATURE_CACHE and pos_perturbed.is_cuda and not self.training
                and ep.E > 0 and _escore.score_supported(self, pl)):
            return None
        static = self._static_features(node_2D_repr, ep)
        if static is None:
            return None
        self.score_network._calls += self.score_network.seed_dev is None
        return _escore.score_nograd(self, ep, pl, static[1], static[0], pos_perturbed)

    @torch.no_grad()
    def get_score_raw(self, node_2D_repr, data, pos_perturbed):
        """The score network's output before the division by -std(t) (get_score = -this / std): the fused sampler kernels
        (msde_pc_corrector / msde_pc_predictor) apply the scaling themselves."""
        pl, ep = self._plan(data)
        out = self._score_fused(node_2D_repr, pos_perturbed, pl, ep)
        if out is not None:
            return out
        node_attr, edge_attr, basis = self._edge_and_node_features(node_2D_repr, pos_perturbed, ep)
        return self.score_network(ep, node_attr, edge_attr, basis, pl)["gradient"]

    @torch.no_grad()
    def get_score(self, node_2D_repr, data, pos_perturbed, sigma, t_pos):
        pl, ep = self._plan(data)
        output = self._score_fused(node_2D_repr, pos_perturbed, pl, ep)
        if output is None:
            node_attr, edge_attr, basis = self._edge_and_node_features(node_2D_repr, pos_perturbed, ep)
            output = self.score_network(ep, node_attr, edge_attr, basis, pl)["gradient"]
        _, std_pos = self.sde_pos.marGINal_prob(pos_perturbed, t_pos)
        return -output / std_pos[:, None]



class SDEModel2Dto3D_01(SDEModel2Dto3D_02):
    """SDE_model_2D_to_3D.py:69-249: `_02` without the distance branch (no `dist_gaussian_fourier` / `input_mlp`,
    edge_attr = edge_attr_2D + edge_attr_3D_frame_invariant, :181).  Same kernels; state-dict keys and order as the
    reference's class.  Several published checkpoints use it (README_checkpoints.md)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        del self.dist_gaussian_fourier
        del self.input_mlp
