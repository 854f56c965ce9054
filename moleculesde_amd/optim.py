"""Flat-buffer Adam on the HIP kernel `msde_adam_flat` + gradient flattening for data parallelism.

Mirrors torch.optim.Adam(param_groups, lr, weight_decay) as used by
examples/pretrain_MoleculeSDE.py:331-337: one lr per model (param group), betas (0.9, 0.999),
eps 1e-8.  All trainable parameters are re-pointed into ONE contiguous fp32 buffer ordered
GIN, SchNet, 2D->3D, 3D->2D, so that (a) the optimiser is a single kernel launch and (b) the
data-parallel gradient exchange is a single RCCL all-reduce of one message.

Difference to torch.optim.Adam, by design: a parameter that received no gradient in a step is
treated as having a zero gradient (torch skips it).  Every parameter on the pretrain path
receives a gradient, so the two coincide there.
"""
import torch

from . import hip


class FlatAdam:
    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        """groups: list of {"params": iterable of Parameters, "lr": float}."""
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.params, seg_end, seg_lr = [], [], []
        seen = set()
        total = 0
        for g in groups:
            for p in g["params"]:
                if not p.requires_grad or id(p) in seen:
                    continue
                seen.add(id(p))
                self.params.append(p)
                total += p.numel()
            seg_end.append(total)
            seg_lr.append(float(g["lr"]))
        dev = self.params[0].device
        self.n = total
        self.flat_p = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad_views = []
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                view = self.flat_p[off:off + n].view_as(p)
                view.copy_(p.data)
                p.data = view                       # parameters now alias the flat buffer
                self.grad_views.append(self.flat_g[off:off + n].view_as(p))
                off += n
        self.seg_end = torch.tensor(seg_end, dtype=torch.int64, device=dev)
        self.seg_lr = torch.tensor(seg_lr, dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def gather_grads(self):
        """Copy the per-parameter .grad tensors into the flat gradient buffer (multi-tensor copy)."""
        have_v, have_g, missing = [], [], []
        for p, gv in zip(self.params, self.grad_views):
            if p.grad is None:
                missing.append(gv)
            else:
                have_v.append(gv)
                have_g.append(p.grad)
        if have_v:
            torch._foreach_copy_(have_v, have_g)
        if missing:
            torch._foreach_zero_(missing)
        return self.flat_g

    def step(self, grad_scale=1.0):
        """Assumes gather_grads() (and, under DP, the all-reduce of flat_g) already happened."""
        self.step_dev.add_(1)
        hip.adam_flat(self.flat_p, self.flat_g, self.m, self.v, self.step_dev, self.seg_end, self.seg_lr,
                      self.betas[0], self.betas[1], self.eps, self.weight_decay, grad_scale)
