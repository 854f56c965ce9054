"""Generate tests/golden/dense_head_prod.npz: EdgeScoreNetwork_dense / NodeScoreNetwork_dense of the GENUINE reference
import (oracle/ref_loader.genuine(): no stand-ins) at the head configuration every MoleculeSDE script uses
(pretrain_MoleculeSDE.py:310-315: nhid = adim = 16, 4 layers, num_linears 3, c_init 2, c_hid 8, c_final 4, 4 heads) --
the configuration the fused kernels of csrc/dense_head.hip are specialised for -- on a small ragged batch: scores, the
gradient of the node features and every parameter gradient of loss = sum(score_edge^2) + sum(score_node^2).

    python oracle/make_golden_dense_prod.py          (needs /root/reference; run in the build container only)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402


def set_parameters(net, base):
    """Deterministic parameter values (a function of the parameter's position only), so that the fixture need not
    store them: tests/test_gpu_models.py fills the product modules the same way.  Biases are non-zero on purpose (the
    reference initialises them to zero, which would hide bias handling)."""
    with torch.no_grad():
        for i, (_, p) in enumerate(net.named_parameters()):
            gen = torch.Generator().manual_seed(base + i)
            if p.dim() == 1:
                p.copy_(torch.rand(p.shape, generator=gen) * 0.6 - 0.3)
            else:
                fan = p.shape[1] if p.shape[0] != p.shape[1] else p.shape[0]
                p.copy_(torch.randn(p.shape, generator=gen) / fan ** 0.5)


def main():
    g = ref_loader.genuine()
    torch.manual_seed(2024)
    B, N, Fd, nout = 5, 9, 8, 11
    # 3 edge layers (2->8, 8->8, 8->4 channels: every layer kind of the 4-layer production net) keep the fixture small
    edge = g.EdgeScoreNetwork_dense(dim3D=Fd, nhid=16, num_layers=3, num_linears=3, c_init=2, c_hid=8, c_final=4, adim=16,
                                    num_heads=4, conv="MLP")
    node = g.NodeScoreNetwork_dense(nfeat=Fd, depth=4, nhid=16, nout=nout)
    set_parameters(edge, 7000)
    set_parameters(node, 9000)
    sizes = [9, 4, 7, 2, 6]
    flags = torch.zeros(B, N)
    for b, n in enumerate(sizes):
        flags[b, :n] = 1
    flags[2, 3] = 0                      # an atom without bonds inside a molecule (node_flags = 0, App. B.5)
    x = torch.randn(B, N, Fd, requires_grad=True)
    a = torch.randn(B, N, N) * 0.7
    a = (a + a.transpose(1, 2)) * 0.5
    a = a * (1 - torch.eye(N))
    a = a * flags[:, :, None] * flags[:, None, :]
    se = edge(x, a, flags)
    sn = node(x, a, flags)
    (se.pow(2).sum() + sn.pow(2).sum()).backward()
    out = dict(x=x.detach().numpy(), adj=a.numpy(), flags=flags.numpy(), sizes=np.array(sizes),
               score_edge=se.detach().numpy(), score_node=sn.detach().numpy(), grad_x=x.grad.numpy())
    out["param_names_edge"] = np.array([k for k, _ in edge.named_parameters()])
    out["param_names_node"] = np.array([k for k, _ in node.named_parameters()])
    for pre, net in (("edge", edge), ("node", node)):
        for k, p in net.named_parameters():
            if p.grad is not None:
                out[f"{pre}.grad.{k}"] = p.grad.numpy().copy()
    path = os.path.join(ROOT, "tests", "golden", "dense_head_prod.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB;", "params without grad:",
          [k for k, p in list(edge.named_parameters()) if p.grad is None][:4], "...")


if __name__ == "__main__":
    main()
