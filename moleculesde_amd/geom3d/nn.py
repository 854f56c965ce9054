"""Small host-side building blocks shared by the model classes (state_dict-compatible with the
reference's modules).  Dense node/edge-level Linear layers are plain library GEMMs (rocBLAS via
torch); everything graph-shaped goes through moleculesde_amd.hip."""
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip, plan as _plan

_LN2 = float(torch.log(torch.tensor(2.0)).item())


def shifted_softplus(x):
    """softplus(x) - ln 2 (reference ShiftedSoftplus, schnet.py:210-216)."""
    return F.softplus(x) - _LN2


import os as _os
FUSED_MLP = True     # Linear/SiLU/Linear chains as hip.mlp_fused (False: layer by layer, the cross-check)


def _need_device(x):
    """The product has no CPU path: a host tensor is an error, not a silent torch fallback."""
    if not x.is_cuda:
        raise hip._lib.MsdeHipError("moleculesde_amd modules need tensors on the HIP device (no CPU fallback)")


class Linear(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) whose forward, input gradient and weight/bias
    gradient run on the fp32 MFMA GEMM of csrc/linear.hip.  `shared`: the module is applied more than once per
    forward, so autograd adds its weight gradients (they are then reduced on the spot, not in the batch)."""

    shared = False

    def forward(self, x):
        _need_device(x)
        return hip.linear(x, self.weight, self.bias, offload=not self.shared)

    def fork(self, x):
        """(x, self(x)) for blocks that also feed x to a residual: see hip.linear_fork."""
        _need_device(x)
        if torch.is_grad_enabled() and x.requires_grad:
            return hip.linear_fork(x, self.weight, self.bias, not self.shared)
        return x, self.forward(x)


class BatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d (same parameters / buffers / state_dict keys).  Training mode runs the two-launch
    HIP kernels of csrc/norm.hip, optionally with the following ReLU fused (`fuse_relu`); eval mode is
    the affine map on the running statistics."""

    fuse_relu = False
    # num_batches_tracked only feeds the momentum=None (cumulative average) mode; with a fixed momentum its per-call
    # increment is a kernel launch that changes no result.  The increments are counted on the host instead
    # (`pending_batches`; a trainer replaying a captured graph adds its replays) and folded into the buffer
    # whenever a state dict is taken, so checkpoints are identical to the reference's.
    pending_batches = 0

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: module.flush_batches_tracked())

    def _load_from_state_dict(self, *a, **k):
        self._nbt_host = None             # the host mirror of num_batches_tracked (momentum=None mode) follows the buffer
        return super()._load_from_state_dict(*a, **k)

    def reset_running_stats(self):
        # (also reached through reset_parameters): the cumulative average restarts at 1 / 1, as torch.nn.BatchNorm1d's does
        self._nbt_host = None
        self.pending_batches = 0
        return super().reset_running_stats()

    def flush_batches_tracked(self):
        if self.pending_batches and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(int(self.pending_batches))
        self.pending_batches = 0

    def forward(self, x):
        _need_device(x)
        if x.dim() != 2:                  # [N, C, L] inputs: not on the hot path, torch's own kernel
            y = super().forward(x)
            return F.relu(y) if self.fuse_relu else y
        if self.training or not self.track_running_stats:
            if self.momentum is None and torch.cuda.is_current_stream_capturing():
                # (checked BEFORE the counter moves: a refused capture attempt must leave num_batches_tracked alone)
                raise RuntimeError("BatchNorm1d(momentum=None) cannot run inside a captured step")
            if self.track_running_stats and self.num_batches_tracked is not None:
                if self.momentum is None:
                    self.flush_batches_tracked()
                    self.num_batches_tracked.add_(1)
                elif not torch.cuda.is_current_stream_capturing():
                    self.pending_batches += 1          # replays of a captured step are counted by the trainer
            momentum = self.momentum
            if momentum is None:
                # cumulative moving average (torch.nn.BatchNorm1d with momentum=None): factor 1 / batches seen.  The count
                # lives on the device; a host mirror is read once and advanced with it (nothing else writes the buffer
                # between forwards except load_state_dict, which resets the mirror below).  The factor changes per call,
                # so this mode cannot be captured into a hipGraph.
                if getattr(self, "_nbt_host", None) is None or self.num_batches_tracked is None:
                    self._nbt_host = int(self.num_batches_tracked) if self.num_batches_tracked is not None else 1
                else:
                    self._nbt_host += 1
                momentum = 1.0 / max(self._nbt_host, 1)
            return hip.batch_norm_train(x, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                                        self.running_var if self.track_running_stats else None, self.eps, momentum,
                                        self.fuse_relu)
        scale = self.weight * torch.rsqrt(self.running_var + self.eps)
        y = x * scale + (self.bias - self.running_mean * scale)
        return F.relu(y) if self.fuse_relu else y


def bn_fusable(bn):
    """Training-mode BatchNorm1d with running statistics and affine parameters: what the fused products implement."""
    return (isinstance(bn, BatchNorm1d) and bn.training and bn.track_running_stats and bn.affine
            and bn.running_mean is not None and bn.momentum is not None)     # (momentum=None: a per-call factor, unfused)


def count_batch(bn):
    """num_batches_tracked bookkeeping of a BatchNorm whose forward ran inside a fused kernel (see BatchNorm1d)."""
    if bn.num_batches_tracked is None:
        return
    if bn.momentum is None:
        bn.flush_batches_tracked()
        bn.num_batches_tracked.add_(1)
    elif not torch.cuda.is_current_stream_capturing():
        bn.pending_batches += 1


def linear(x, weight, bias=None):
    _need_device(x)
    return hip.linear(x, weight, bias)


def linear_fork(x, weight, bias=None):
    _need_device(x)
    if torch.is_grad_enabled() and x.requires_grad:
        return hip.linear_fork(x, weight, bias)
    return x, linear(x, weight, bias)


class ShiftedSoftplus(nn.Module):
    def forward(self, x):
        return shifted_softplus(x)


class MultiLayerPerceptron(nn.Module):
    """Same parameters/keys as layers/common.py:5-40 (`layers.{i}.weight|bias`, Xavier / zero init)."""

    def __init__(self, input_dim, hidden_dims, activation="relu", dropout=0):
        super().__init__()
        self.dims = [input_dim] + list(hidden_dims)
        self.activation = getattr(F, activation) if isinstance(activation, str) else None
        self.activation_name = activation if isinstance(activation, str) else None
        self.dropout = nn.Dropout(dropout) if dropout else None
        self.layers = nn.ModuleList([Linear(self.dims[i], self.dims[i + 1]) for i in range(len(self.dims) - 1)])
        self.reset_parameters()

    def reset_parameters(self):
        for layer in self.layers:
            nn.init.xavier_uniform_(layer.weight)
            nn.init.constant_(layer.bias, 0.0)

    def forward(self, x):
        if (FUSED_MLP and len(self.layers) >= 2 and self.activation_name == "silu" and not self.dropout
                and x.is_cuda and x.dim() == 2 and all(d % 4 == 0 for d in self.dims) and not any(l.shared for l in self.layers)):
            # bias + SiLU in the GEMM epilogues, SiLU' in the epilogue of the next layer's input-gradient GEMM
            return hip.mlp_fused(x, [(l.weight, l.bias) for l in self.layers], "silu")
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if i < len(self.layers) - 1:
                if self.activation:
                    x = self.activation(x)
                if self.dropout:
                    x = self.dropout(x)
        return x


class EmbeddingList(nn.Module):
    """ogb AtomEncoder / BondEncoder parameters (`<name>.{k}.weight`, Xavier-uniform)."""

    def __init__(self, dims, emb_dim, list_name):
        super().__init__()
        lst = nn.ModuleList()
        for d in dims:
            e = nn.Embedding(d, emb_dim)
            nn.init.xavier_uniform_(e.weight.data)
            lst.append(e)
        setattr(self, list_name, lst)
        self._list_name = list_name

    def table(self):
        """Concatenated [sum(dims), D] table, the layout the kernels index with pre-offset codes."""
        ws = [e.weight for e in getattr(self, self._list_name)]
        return hip.cat_params(ws)

    def fusion_sets(self):
        """Parameters that want to be adjacent in the optimiser's flat buffer (then table() is a free view)."""
        return [[e.weight for e in getattr(self, self._list_name)]]


# ---- plan registry: lets the reference-style call signatures (tensors, not a Batch) find the plan --
_REGISTRY = {}


def register_plan(data, pl):
    for name in ("x", "batch", "edge_index"):
        t = getattr(data, name, None)
        if isinstance(t, torch.Tensor):
            _REGISTRY[id(t)] = (weakref.ref(t), pl)
    if len(_REGISTRY) > 64:
        for k in [k for k, (r, _) in _REGISTRY.items() if r() is None]:
            _REGISTRY.pop(k, None)


def lookup_plan(t):
    ent = _REGISTRY.get(id(t))
    if ent is not None and ent[0]() is t:
        return ent[1]
    return None


def prepare_batch(data, device=None, **kw):
    """Collation-time hook of the drivers: build the index plan on the host, move everything to the
    device, and register the plan so `GNN(batch.x, batch.edge_index, batch.edge_attr)` and
    `SchNet(batch.x[:, 0], batch.positions, batch.batch)` (the reference's call forms,
    pretrain_MoleculeSDE.py:128-131) find it."""
    _plan.prepare_batch(data, device=device, **kw)
    register_plan(data, data._msde_plan)
    return data


class DeviceNoise:
    """Default noise source: device RNG (torch's HIP generator)."""

    replay = False     # draws are not tied to a program order: independent branches may run concurrently
    seed, calls, seed_dev = 0x5EED, 0, None     # class defaults: subclasses need not call __init__

    def __init__(self, seed=0x5EED):
        self.seed = int(seed)
        self.calls = 0          # host-side draw counter (eager); under hipGraph replay `seed_dev` varies the draws
        self.seed_dev = None    # device uint64 step counter, set by the trainer

    def randn_like(self, x):
        return torch.randn_like(x)

    def randint(self, high, size, device):
        return torch.randint(0, high, size=size, device=device)

    def randperm(self, n, device):
        if torch.device(device).type == "cuda" and n <= hip.RANDPERM_MAX:
            self.calls += 1
            return hip.randperm(n, device, self.seed + 0x9E3779B1 * self.calls, self.seed_dev)
        return torch.randperm(n, device=device)

    def draws_in_kernel(self):
        """True when nothing overrides the position-noise / time-step draws: the VE perturbation kernel may then make them
        itself (counter-based generator) instead of reading torch.randn_like / torch.randint results."""
        return type(self).randn_like is DeviceNoise.randn_like and type(self).randint is DeviceNoise.randint

    def next_seed(self):
        self.calls += 1
        return self.seed + 0x9E3779B1 * self.calls

    def rand(self, n, device):
        return torch.rand(n, device=device)

    def randperm_pair(self, n, device):
        """The two permutations of dual_CL from one launch (a subclass that overrides randperm is asked twice)."""
        if type(self).randperm is not DeviceNoise.randperm:
            return self.randperm(n, device), self.randperm(n, device)
        if torch.device(device).type == "cuda" and n <= hip.RANDPERM_MAX:
            self.calls += 1
            p = hip.randperm(n, device, self.seed + 0x9E3779B1 * self.calls, self.seed_dev, count=2)
            return p[0], p[1]
        return self.randperm(n, device), self.randperm(n, device)


class CpuReplayNoise:
    """Parity noise source: draws from a torch CPU generator in the reference's program order
     and uploads, so a reference/oracle run under torch.manual_seed(seed)
    sees the same numbers."""

    replay = True      # draws must happen in the reference's program order

    def __init__(self, seed):
        self.g = torch.Generator(device="cpu")
        self.g.manual_seed(int(seed))

    def randperm_pair(self, n, device):
        return self.randperm(n, device), self.randperm(n, device)     # the reference's two consecutive draws

    def randn_like(self, x):
        return torch.randn(x.shape, generator=self.g, dtype=x.dtype).to(x.device)

    def randint(self, high, size, device):
        return torch.randint(0, high, size=size, generator=self.g).to(device)

    def randperm(self, n, device):
        return torch.randperm(n, generator=self.g).to(device)

    def rand(self, n, device):
        return torch.rand(n, generator=self.g).to(device)
