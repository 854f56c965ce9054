"""Generate tests/golden/f4_painn.npz from the reference's own PaiNN (Geom3D/models/painn.py:118-269 on the stand-in
third-party layer; needs /root/reference): a 4-molecule batch with a radius graph, 3 interaction blocks; stored are the
readout, the latent atom features, the gradient of loss = sum(h^2) + sum(q) w.r.t. the positions (the MD17 force path
differentiates the same graph) and w.r.t. every parameter.  Parameters are a deterministic function of their position
(make_golden_dense_prod.set_parameters, repeated by the test) so the fixture need not store them.
    python oracle/make_golden_painn.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402
from oracle.make_golden import batch_np, grads_np  # noqa: E402
from oracle.make_golden_dense_prod import set_parameters  # noqa: E402
from moleculesde_amd.synthetic import make_batch  # noqa: E402


def main():
    ns = ref_loader.verbatim()
    import importlib
    PaiNN = importlib.import_module("Geom3D.models").PaiNN
    radius_graph = ns.standins.nn.radius_graph
    b = make_batch(4, seed=23, sizes=[7, 4, 9, 5])
    cutoff = 3.5
    ei = radius_graph(b.positions, r=cutoff, batch=b.batch, loop=False)
    torch.manual_seed(5)
    m = PaiNN(n_atom_basis=32, n_interactions=3, n_rbf=20, cutoff=cutoff, max_z=119, n_out=1, readout="mean")
    set_parameters(m, 6100)
    with torch.no_grad():
        m.embedding.weight[0].zero_()              # padding_idx = 0 (painn.py:172): the row is zero and stays zero
    pos = b.positions.clone().requires_grad_(True)
    h, q = m(b.x[:, 0], pos, ei, b.batch, return_latent=True)
    loss = h.pow(2).sum() + q.sum()
    loss.backward()
    out = dict(radius_edge_index=ei.numpy(), cutoff=np.float32(cutoff), h=h.detach().numpy(), q=q.detach().numpy(),
               grad_pos=pos.grad.numpy(), param_names=np.array([k for k, _ in m.named_parameters()]),
               n_edges=np.int64(ei.size(1)))
    out.update(batch_np(b))
    out.update(grads_np(m))
    path = os.path.join(ROOT, "tests", "golden", "f4_painn.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB; edges", ei.size(1), "atoms", b.x.size(0),
          "h[0,:3]", h[0, :3].tolist())


if __name__ == "__main__":
    main()
