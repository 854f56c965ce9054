"""SDEModel3Dto2D_node_adj_dense + the dense edge/node score networks.

API, constructor arguments, `forward(node_3D_repr, data, continuous, train, reduce_mean,
anneal_power) -> (loss_x, loss_adj)` and state_dict keys mirror
SDE_model_3D_to_2D_node_adj_dense.py:13-179, invariant_scorenetwork_dense.py:40-131,
layers/edge_network_dense.py:33-128, layers/node_network_dense.py:25-85.

Device mapping (round 1): the ragged -> padded packing uses the row-gather kernel and a plan built at
collation; every Linear of the MLP chains (incl. the 364 -> 728 -> 728 -> 119 node head, the real GEMM
of this model) goes through `hip.linear` (fp32 MFMA kernels / vendor GEMM per the dispatch policy);
the 20 x 20 per-molecule products are batched library GEMMs.  The C_in parallel `EdgeLayer`s of one
`EdgeNetwork_dense` share their input, so they are evaluated as ONE stacked problem (first layers
concatenated along the output dim, later layers as a batched GEMM over channels) instead of C_in
separate module calls -- same arithmetic per channel, ~10x fewer launches.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, hip, plan as _plan
from . import dense_head as _dh
from . import nn as _nn
from .sde import VESDE, VPSDE

USE_FUSED_HEAD = True     # False: the operator-by-operator path below (cross-check; shapes the kernels do not cover)

EPSILON = 1e-6


def mask_x(x, flags):
    return x * flags[:, :, None]


def mask_adjs(adjs, flags):
    if adjs.dim() == 4:
        flags = flags.unsqueeze(1)
    return adjs * flags.unsqueeze(-1) * flags.unsqueeze(-2)


class NodeNetwork_dense(nn.Module):
    """Dense GCN layer (layers/node_network_dense.py:25-85): weight stored [in, out], glorot/zero init."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.Tensor(in_channels, out_channels))
        self.bias = nn.Parameter(torch.Tensor(out_channels))
        stdv = math.sqrt(6.0 / (in_channels + out_channels))
        self.weight.data.uniform_(-stdv, stdv)
        self.bias.data.fill_(0)


def _norm_adj(adj):
    """Self loops set to 1, symmetric degree normalisation with clamp(min=1) (:66-74)."""
    n = adj.size(-1)
    eye = torch.eye(n, device=adj.device, dtype=adj.dtype)
    adj = adj * (1 - eye) + eye
    dis = adj.sum(dim=-1).clamp(min=1).pow(-0.5)
    return dis.unsqueeze(-1) * adj * dis.unsqueeze(-2)


class EdgeLayer(nn.Module):
    """Parameters of one EdgeLayer (conv='MLP'); evaluated in a stacked fashion by EdgeNetwork_dense."""

    def __init__(self, in_dim, attn_dim, out_dim, num_heads, conv):
        super().__init__()
        if conv != "MLP":
            raise NotImplementedError("conv='GCN' is never selected on the MoleculeSDE path")
        self.num_heads, self.attn_dim, self.out_dim = num_heads, attn_dim, out_dim
        self.func_q = _nn.MultiLayerPerceptron(in_dim, [2 * attn_dim, 2 * attn_dim], activation="tanh")
        self.func_k = _nn.MultiLayerPerceptron(in_dim, [2 * attn_dim, 2 * attn_dim], activation="tanh")
        self.func_v = NodeNetwork_dense(in_dim, out_dim)


class EdgeNetwork_dense(nn.Module):
    def __init__(self, num_linears, conv_input_dim, attn_dim, conv_output_dim, input_dim, output_dim, num_heads, conv):
        super().__init__()
        self.attn_dim, self.num_heads = attn_dim, num_heads
        self.attn = nn.ModuleList([EdgeLayer(conv_input_dim, attn_dim, conv_output_dim, num_heads, conv)
                                   for _ in range(input_dim)])
        self.hidden_dim = 2 * max(input_dim, output_dim)
        self.mlp = _nn.MultiLayerPerceptron(2 * input_dim, [self.hidden_dim] * (num_linears - 1) + [output_dim],
                                            activation="elu")
        self.multi_channel = _nn.MultiLayerPerceptron(input_dim * conv_output_dim, [self.hidden_dim, conv_output_dim],
                                                      activation="elu")

    def fusion_sets(self):
        """Parameters the fused kernels consume as ONE stacked operand (dense_head.edge_layer_tensors): laid out back
        to back by FlatAdam so that the stacks are free views."""
        at = list(self.attn)
        return [[a.func_q.layers[0].weight for a in at] + [a.func_k.layers[0].weight for a in at],
                [a.func_q.layers[0].bias for a in at] + [a.func_k.layers[0].bias for a in at],
                [a.func_q.layers[1].weight for a in at] + [a.func_k.layers[1].weight for a in at],
                [a.func_q.layers[1].bias for a in at] + [a.func_k.layers[1].bias for a in at],
                [a.func_v.weight for a in at], [a.func_v.bias for a in at]]

    def _stacked_mlp(self, x2, which):
        """func_q / func_k of all C channels at once: x2 [BN, F] -> [C, BN, 2*attn]."""
        C = len(self.attn)
        mods = [getattr(a, which) for a in self.attn]
        W0 = torch.cat([m.layers[0].weight for m in mods], dim=0)          # [C*H, F]
        b0 = torch.cat([m.layers[0].bias for m in mods], dim=0)
        H = mods[0].layers[0].weight.size(0)
        h = torch.tanh(_nn.linear(x2, W0, b0))                              # [BN, C*H]
        h = h.view(-1, C, H).transpose(0, 1)                                # [C, BN, H]
        W1 = torch.stack([m.layers[1].weight for m in mods])               # [C, H, H]
        b1 = torch.stack([m.layers[1].bias for m in mods]).unsqueeze(1)
        return torch.baddbmm(b1, h, W1.transpose(1, 2))                     # last layer: no activation

    def forward(self, x, adj, flags):
        """x [B,N,F], adj [B,C,N,N] -> x_out [B,N,F_o], adj_out [B,C_o,N,N] (:105-128)."""
        B, N, Fd = x.shape
        C = len(self.attn)
        x2 = x.reshape(B * N, Fd)
        Q = self._stacked_mlp(x2, "func_q")                                 # [C, BN, 2a]
        K = self._stacked_mlp(x2, "func_k")
        # V_c = A_norm_c (x W_c) + b_c  (NodeNetwork_dense.forward)
        Wv = torch.cat([a.func_v.weight for a in self.attn], dim=1)         # [F, C*out]
        bv = torch.stack([a.func_v.bias for a in self.attn])                # [C, out]
        out_dim = self.attn[0].out_dim
        xw = torch.mm(x2, Wv).view(B, N, C, out_dim).permute(2, 0, 1, 3)    # [C, B, N, out]
        an = _norm_adj(adj).transpose(0, 1)                                 # [C, B, N, N]
        V = torch.matmul(an, xw) + bv.view(C, 1, 1, out_dim)                # [C, B, N, out]
        # attention: split 2a into chunks of attn_dim // num_heads (8 effective heads, App. B.3)
        ds = self.attn_dim // self.num_heads
        nh = Q.size(-1) // ds
        Qh = Q.view(C, B, N, nh, ds).permute(0, 1, 3, 2, 4)                 # [C, B, nh, N, ds]
        Kh = K.view(C, B, N, nh, ds).permute(0, 1, 3, 2, 4)
        A = torch.tanh(torch.matmul(Qh, Kh.transpose(-1, -2)) / math.sqrt(ds)).mean(dim=2)   # [C, B, N, N]
        A = (A + A.transpose(-1, -2)) / 2
        # node update
        xcat = V.permute(1, 2, 0, 3).reshape(B, N, C * out_dim)             # cat over channels, channel-major
        x_out = torch.tanh(mask_x(self.multi_channel(xcat), flags))
        # edge update: mlp over [mask channels, adjacency channels]
        mlp_in = torch.cat([A.permute(1, 2, 3, 0), adj.permute(0, 2, 3, 1)], dim=-1)          # [B, N, N, 2C]
        mlp_out = self.mlp(mlp_in.reshape(-1, 2 * C))
        _adj = mlp_out.view(B, N, N, -1).permute(0, 3, 1, 2)
        _adj = _adj + _adj.transpose(-1, -2)
        return x_out, mask_adjs(_adj, flags)


class EdgeScoreNetwork_dense(nn.Module):
    def __init__(self, dim3D, nhid, num_layers, num_linears, c_init, c_hid, c_final, adim, num_heads, conv):
        super().__init__()
        self.c_init, self.num_layers = c_init, num_layers
        self.layers = nn.ModuleList()
        for l in range(num_layers):
            if l == 0:
                self.layers.append(EdgeNetwork_dense(num_linears, dim3D, nhid, nhid, c_init, c_hid, num_heads, conv))
            elif l == num_layers - 1:
                self.layers.append(EdgeNetwork_dense(num_linears, nhid, adim, nhid, c_hid, c_final, num_heads, conv))
            else:
                self.layers.append(EdgeNetwork_dense(num_linears, nhid, adim, nhid, c_hid, c_hid, num_heads, conv))
        self.fdim = c_hid * (num_layers - 1) + c_final + c_init
        self.final = _nn.MultiLayerPerceptron(self.fdim, [2 * self.fdim, 2 * self.fdim, 1], activation="silu")

    def forward(self, x, adj, flags):
        chans = [adj]
        a = adj
        for _ in range(self.c_init - 1):                                    # pow_tensor (:28-37)
            a = torch.bmm(a, adj)
            chans.append(a)
        adjc = torch.stack(chans, dim=1)
        adj_list = [adjc]
        for layer in self.layers:
            x, adjc = layer(x, adjc, flags)
            adj_list.append(adjc)
        adjs = torch.cat(adj_list, dim=1).permute(0, 2, 3, 1)               # [B, N, N, fdim]
        B, N = adjs.shape[:2]
        score = self.final(adjs.reshape(-1, self.fdim)).view(B, N, N)
        score = score * (1.0 - torch.eye(N, device=score.device, dtype=score.dtype)).unsqueeze(0)
        return mask_adjs(score, flags)


class NodeScoreNetwork_dense(nn.Module):
    def __init__(self, nfeat, depth, nhid, nout):
        super().__init__()
        self.nfeat, self.depth, self.nhid, self.nout = nfeat, depth, nhid, nout
        self.layers = nn.ModuleList([NodeNetwork_dense(nfeat if l == 0 else nhid, nhid) for l in range(depth)])
        self.fdim = nfeat + depth * nhid
        self.final = _nn.MultiLayerPerceptron(self.fdim, [2 * self.fdim, 2 * self.fdim, nout], activation="silu")

    def fusion_sets(self):
        return [[l.weight for l in self.layers[1:]], [l.bias for l in self.layers]]

    def forward(self, x, adj, flags):
        B, N, _ = x.shape
        an = _norm_adj(adj)
        x_list = [x]
        for layer in self.layers:
            x = torch.tanh(torch.matmul(an, torch.matmul(x, layer.weight)) + layer.bias)
            x_list.append(x)
        xs = torch.cat(x_list, dim=-1).reshape(B * N, self.fdim)
        out = self.final(xs).view(B, N, -1)                                  # 364 -> 728 -> 728 -> 119 GEMM chain
        return mask_x(out, flags)


class SDEModel3Dto2D_node_adj_dense(nn.Module):
    CONCAT_EMBEDDINGS = False     # _02 (below): the two embeddings are concatenated instead of added

    def __init__(self, dim3D, nhid, num_layers, num_linears, c_hid, c_final, adim, emb_dim, beta_min, beta_max,
                 num_diffusion_timesteps, c_init=1, num_heads=4, conv="MLP", noise_mode="discrete", SDE_type="VE",
                 num_class_X=119, noise_on_one_hot=True):
        super().__init__()
        self.emb_dim, self.beta_min, self.beta_max = emb_dim, beta_min, beta_max
        self.num_diffusion_timesteps, self.nfeat, self.nhid = num_diffusion_timesteps, dim3D, nhid
        self.num_layers, self.num_linears, self.c_init, self.c_hid, self.c_final = num_layers, num_linears, c_init, c_hid, c_final
        self.adim, self.num_heads, self.conv, self.noise_mode, self.SDE_type = adim, num_heads, conv, noise_mode, SDE_type
        if SDE_type == "VE":
            self.sde_x = VESDE(beta_min, beta_max, num_diffusion_timesteps)
            self.sde_adj = VESDE(beta_min, beta_max, num_diffusion_timesteps)
        elif SDE_type == "VP":
            self.sde_x = VPSDE(beta_min, beta_max, num_diffusion_timesteps)
            self.sde_adj = VPSDE(beta_min, beta_max, num_diffusion_timesteps)
        else:
            raise NotImplementedError(SDE_type)
        self.num_class_X, self.noise_on_one_hot = num_class_X, noise_on_one_hot
        self.embedding_X = _nn.Linear(num_class_X if noise_on_one_hot else 1, dim3D)
        self.embedding_3D = _nn.Linear(dim3D, dim3D)
        net_in = 2 * dim3D if self.CONCAT_EMBEDDINGS else dim3D
        self.edge_score_network = EdgeScoreNetwork_dense(dim3D=net_in, nhid=nhid, num_layers=num_layers,
                                                         num_linears=num_linears, c_init=c_init, c_hid=c_hid,
                                                         c_final=c_final, adim=adim, num_heads=4, conv=conv)
        self.node_score_network = NodeScoreNetwork_dense(nfeat=net_in, depth=num_layers, nhid=nhid,
                                                         nout=num_class_X if noise_on_one_hot else 1)
        self.noise = _nn.DeviceNoise()

    def _forward_fused(self, node_3D_repr, data, reduce_mean, anneal_power, pl, dn):
        """Product path: one autograd node, only libmsde_hip kernels (geom3d/dense_head.py)."""
        device = node_3D_repr.device
        B, Nm, T = pl.B, dn.N_max, self.num_diffusion_timesteps
        cfg = getattr(dn, "_fused_cfg", None)
        if cfg is None:
            import types
            cfg = types.SimpleNamespace(
                N=pl.N, P=dn.P, B=B, n_max=Nm, mol_ptr=pl.mol_ptr, pair_ptr=dn.pair_ptr, bond_rowptr=pl.bond.rowptr,
                bond_src=pl.bond.src, bond_val=pl.bond_type, z_atom=pl.z_codes.view(-1), T=T, eps=EPSILON,
                ncls=self.num_class_X)
            cfg.chans, cfg.offs = _dh.edge_net_shape(self.edge_score_network)
            dn._fused_cfg = cfg
        cfg.sde_vp = 1 if self.SDE_type == "VP" else 0
        cfg.concat = bool(self.CONCAT_EMBEDDINGS)          # _02: embeddings concatenated (:326), 2 * dim3D-wide score networks
        cfg.p0, cfg.p1 = float(self.beta_min), float(self.beta_max)
        cfg.draws = cfg.t_in = cfg.noise_adj = cfg.noise_x = None
        cfg.nm_pad, cfg.seed, cfg.seed_dev = Nm, 0, None
        noise = self.noise
        if getattr(noise, "replay", False) or type(noise).randn_like is not _nn.DeviceNoise.randn_like:
            # replayable noise source: the reference's draws in its program order (:112, :135, :144), padded shapes
            if self.noise_mode == "discrete":
                cfg.draws = noise.randint(T, (B // 2 + 1,), device).contiguous()
            else:
                cfg.t_in = (noise.rand(B, device) * (1 - EPSILON) + EPSILON).float().contiguous()
            cfg.noise_adj = noise.randn_like(torch.empty(B, Nm, Nm, device=device)).contiguous()
            cfg.noise_x = noise.randn_like(torch.empty(B, Nm, self.num_class_X, device=device)).contiguous()
        else:
            # device noise: counter-based draws inside the prepare kernel (no operator launches at all)
            noise.calls += 1
            cfg.seed = (noise.seed * 0x9E3779B1 + 0x3D2D * noise.calls) & 0xFFFFFFFFFFFFFFFF
            cfg.seed_dev = noise.seed_dev
            if self.noise_mode != "discrete":
                cfg.t_in = (noise.rand(B, device) * (1 - EPSILON) + EPSILON).float().contiguous()
        cfg.anneal = float(anneal_power)
        # capacity buckets (moleculesde_amd.bucket): N_max of the loaded batch lives on the device
        cfg.nmax_dev = getattr(dn, "nmax_dev", None) if reduce_mean else None
        if reduce_mean:
            cfg.scale_x, cfg.scale_adj = 1.0 / (B * Nm * self.num_class_X), 1.0 / (B * Nm * Nm)
        else:
            cfg.scale_x = cfg.scale_adj = 0.5 / B
        TE = _dh.edge_net_tensors(self.edge_score_network)
        TN = _dh.node_net_tensors(self.node_score_network)
        cfg.n_edge_tensors = len(TE)
        T4 = [self.embedding_3D.weight, self.embedding_3D.bias, self.embedding_X.weight, self.embedding_X.bias]
        return _dh.dense_head_losses(cfg, node_3D_repr, T4 + TE + TN)

    def forward(self, node_3D_repr, data, continuous, train, reduce_mean, anneal_power):
        if not continuous:
            raise NotImplementedError("Discrete not supported")              # as the reference (:82,92)
        device = node_3D_repr.device
        pl = _plan.get_plan(data)
        dn = _plan.dense_plan(pl, data)                                      # padded layout, built once per batch
        fused_ok = self.noise_on_one_hot and hasattr(pl, "bond_type") and \
            _dh.fused_supported(self.edge_score_network, self.node_score_network, dn.N_max)
        if USE_FUSED_HEAD and fused_ok:
            return self._forward_fused(node_3D_repr, data, reduce_mean, anneal_power, pl, dn)
        if not getattr(self, "allow_operator_path", False) and USE_FUSED_HEAD:
            # the operator path below runs its batched products on the vendor GEMM (torch.baddbmm / matmul): never silently
            raise _lib.MsdeHipError(
                "SDEModel3Dto2D_node_adj_dense: this configuration is outside the fused HIP head (nhid = adim = 16, num_linears = 3, "
                "c_init = 2, 4 layers, <= 32 atoms per molecule, noise_on_one_hot, bond features) -- set "
                "`model.allow_operator_path = True` to run it operator by operator (vendor batched GEMMs)")
        B, Nm, T = pl.B, dn.N_max, self.num_diffusion_timesteps
        if self.noise_mode == "discrete":
            t = self.noise.randint(T, (B // 2 + 1,), device)
            t = torch.cat([t, T - t - 1], dim=0)[:B]
            t = t / T * (1 - EPSILON) + EPSILON
        else:
            t = self.noise.rand(B, device) * (1 - EPSILON) + EPSILON
        adj, flags = dn.adj, dn.flags                                        # (:121-134) data only, no parameters
        h3 = hip.gather_rows_grad(node_3D_repr, dn.pad_idx, dn.node_slot).view(B, Nm, -1)   # to_dense_batch (:130)

        z_adj = self.noise.randn_like(adj).triu(1)                            # gen_noise(sym=True) (:532-540)
        z_adj = mask_adjs(z_adj + z_adj.transpose(-1, -2), flags)
        mean_adj, std_adj = self.sde_adj.marGINal_prob(adj, t)
        perturbed_adj = mask_adjs(mean_adj + std_adj[:, None, None] * z_adj, flags)

        if self.noise_on_one_hot:
            x0 = F.one_hot(dn.z, self.num_class_X).float()                   # (:143)
        else:
            x0 = dn.z.float().unsqueeze(2)
        z_x = mask_x(self.noise.randn_like(x0), flags)
        mean_x, std_x = self.sde_x.marGINal_prob(x0, t)
        perturbed_x = mask_x(mean_x + std_x[:, None, None] * z_x, flags)
        if self.CONCAT_EMBEDDINGS:
            perturbed_x = torch.cat([self.embedding_3D(h3), self.embedding_X(perturbed_x)], -1)   # _02 (:326)
        else:
            perturbed_x = self.embedding_3D(h3) + self.embedding_X(perturbed_x)   # (:156)

        score_adj = -self.edge_score_network(perturbed_x, perturbed_adj, flags) / std_adj[:, None, None]   # (:86-94)
        score_x = -self.node_score_network(perturbed_x, perturbed_adj, flags) / std_x[:, None, None]

        losses_x = torch.square(score_x + z_x)
        losses_adj = torch.square(score_adj + z_adj)
        if anneal_power != 0:
            losses_x = losses_x * (std_x ** anneal_power)[:, None, None]
            losses_adj = losses_adj * (std_adj ** anneal_power)[:, None, None]
        if reduce_mean:
            losses_x = losses_x.reshape(B, -1).mean(dim=-1)
            losses_adj = losses_adj.reshape(B, -1).mean(dim=-1)
        else:
            losses_x = 0.5 * losses_x.reshape(B, -1).sum(dim=-1)
            losses_adj = 0.5 * losses_adj.reshape(B, -1).sum(dim=-1)
        return torch.mean(losses_x), torch.mean(losses_adj)


class SDEModel3Dto2D_node_adj_dense_02(SDEModel3Dto2D_node_adj_dense):
    """SDE_model_3D_to_2D_node_adj_dense.py:182-345: identical to the model above except that embedding_3D(h) and
    embedding_X(x) are CONCATENATED (:326) and both score networks take 2 * dim3D features (:223,233).  Same state_dict
    keys.  Runs on the same fused head (geom3d/dense_head.py): the second embedding product writes the next column block of the
    node-feature buffer instead of accumulating into the first."""
    CONCAT_EMBEDDINGS = True


def build_from_args(args, node_class=119):
    """pretrain_MoleculeSDE.py:276-315 (SDEModel3Dto2D_node_adj_dense branches)."""
    classes = {"SDEModel3Dto2D_node_adj_dense": SDEModel3Dto2D_node_adj_dense,
               "SDEModel3Dto2D_node_adj_dense_02": SDEModel3Dto2D_node_adj_dense_02}
    if args.SDE_3Dto2D_model not in classes:
        raise NotImplementedError(args.SDE_3Dto2D_model)
    ranges = {"VE": ("VE", 0.1, 1.0), "VP": ("VP", 0.2, 1.0), "VE02": ("VE", 0.1, 10.0), "VP02": ("VP", 0.1, 30.0),
              "VE03": ("VE", 0.1, 1000.0), "VP03": ("VP", 0.1, 1000.0)}
    sde_type, bmin, bmax = ranges[args.SDE_type_3Dto2D]
    return classes[args.SDE_3Dto2D_model](
        dim3D=args.emb_dim, c_init=2, c_hid=8, c_final=4, num_heads=4, adim=16, nhid=16, num_layers=4,
        emb_dim=args.emb_dim, num_linears=3, beta_min=bmin, beta_max=bmax, num_diffusion_timesteps=1000,
        SDE_type=sde_type, num_class_X=node_class, noise_on_one_hot=args.noise_on_one_hot)
