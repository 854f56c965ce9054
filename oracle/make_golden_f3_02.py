"""Generate tests/golden/f3_sde3d2d_02.npz from the reference's own files (verbatim on the stand-in third-party layer;
needs /root/reference): SDEModel3Dto2D_node_adj_dense_02 (SDE_model_3D_to_2D_node_adj_dense.py:182-349), the variant that
CONCATENATES embedding_3D(h) and embedding_X(x) (:326) and feeds 2 * dim3D features to both score networks (:223,233).
Parameters are a deterministic function of their position (make_golden_dense_prod.set_parameters, repeated by the test)
so that the fixture need not store them.  Stored: parameter names, the two losses under torch.manual_seed program-order noise, the gradient of (loss_x + loss_adj) w.r.t.
the 3D representation and every parameter, and the input the score networks saw (captured with a forward hook).
    python oracle/make_golden_f3_02.py"""
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
    mod = ns.sde3d2d
    E = 8
    b = make_batch(4, seed=11, sizes=[6, 3, 5, 4])
    torch.manual_seed(91)
    # the score-network dimensions of every MoleculeSDE script (pretrain_MoleculeSDE.py:310-315: nhid = adim = 16, 4 layers,
    # 3 linears, channels 2 | 8 | 8 | 8 -> 4): the configuration the fused head kernels implement, so the test runs THEM
    m = mod.SDEModel3Dto2D_node_adj_dense_02(dim3D=E, c_init=2, c_hid=8, c_final=4, num_heads=4, adim=16, nhid=16,
                                             num_layers=4, emb_dim=E, num_linears=3, beta_min=0.1, beta_max=1.0,
                                             num_diffusion_timesteps=1000, SDE_type="VE", num_class_X=119,
                                             noise_on_one_hot=True)
    set_parameters(m, 4200)
    h3 = torch.randn(b.x.size(0), E, requires_grad=True)
    cap = {}
    m.edge_score_network.register_forward_hook(lambda mod_, inp, out: cap.__setitem__("edge_in", inp[0]))
    seed = 92
    torch.manual_seed(seed)
    lx, la = m(h3, b.clone(), continuous=True, train=True, reduce_mean=True, anneal_power=0)
    assert cap["edge_in"].shape[-1] == 2 * E
    (lx + la).backward()
    out = dict(seed=np.int64(seed), h3=h3.detach().numpy(), loss_x=lx.detach().numpy(), loss_adj=la.detach().numpy(),
               grad_h3=h3.grad.numpy(), net_in=cap["edge_in"].detach().numpy())
    out.update(batch_np(b))
    out["param_names"] = np.array([k for k, _ in m.named_parameters()])
    out.update(grads_np(m))
    path = os.path.join(ROOT, "tests", "golden", "f3_sde3d2d_02.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
