"""Generate tests/golden/f3_variants.npz from the reference's own files (verbatim on the stand-in third-party layer,
oracle/ref_loader.verbatim(); needs /root/reference): the §8-f3 variants of the 2D->3D model --
  (a) SDEModel2Dto3D_01 (VE): loss, parameter gradients, get_score on a 3-molecule toy batch;
  (b) SDEModel2Dto3D_02 with SDE_type='VP': loss, gradients, get_score, and 3 reverse-diffusion predictor steps driven
      by the genuine VPSDE.reverse(model).discretize (SDE_sparse.py:64-102,152-160).
Dropout off, noise = torch.manual_seed program order (replayed by CpuReplayNoise on the product side).
    python oracle/make_golden_f3.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402
from oracle.make_golden import batch_np, disable_dropout, grads_np, sd_np  # noqa: E402
from moleculesde_amd.synthetic import make_batch  # noqa: E402


def main():
    ns = ref_loader.verbatim()
    E = 16
    out = {}
    b = make_batch(3, seed=7, sizes=[5, 4, 6])
    out.update(batch_np(b))
    torch.manual_seed(71)
    h2 = torch.randn(b.x.size(0), E)
    out["h2"] = h2.numpy()
    for tag, cls, sde_type, seed in (("m01", ns.sde2d3d.SDEModel2Dto3D_01, "VE", 301), ("vp", ns.SDEModel2Dto3D_02, "VP", 302)):
        torch.manual_seed(seed)
        m = disable_dropout(cls(emb_dim=E, hidden_dim=32, beta_min=0.2, beta_max=1.0, num_diffusion_timesteps=1000,
                                beta_schedule=None, SDE_type=sde_type, use_extend_graph=True))
        out.update(sd_np(m, f"{tag}.sd."))
        h = h2.clone().requires_grad_(True)
        torch.manual_seed(seed + 10)
        loss = m(h, b.clone(), anneal_power=0)["position"]
        loss.backward()
        out[f"{tag}.loss"] = loss.detach().numpy()
        out[f"{tag}.grad_h2"] = h.grad.numpy()
        out.update(grads_np(m, f"{tag}.grad."))
        out[f"{tag}.seed"] = np.int64(seed + 10)
        m.eval()
        pos = b.positions.detach() * 0.9 + 0.05
        t = torch.full((b.x.size(0),), 0.37)
        out[f"{tag}.score_pos"], out[f"{tag}.score_t"] = pos.numpy(), t.numpy()
        out[f"{tag}.score"] = m.get_score(h2, b.clone(), pos, None, t).numpy()
        if sde_type == "VP":
            rsde = m.sde_pos.reverse(m, probability_flow=False)
            x = pos.clone()
            traj, noises = [], []
            for tv in (0.9, 0.6, 0.2):
                vt = torch.full((b.x.size(0),), tv)
                f, G = rsde.discretize(x, h2, b.clone(), vt)
                z = torch.randn_like(x)
                x = (x - f) + G[:, None] * z
                traj.append(x.clone()); noises.append(z)
            out["vp.pred_ts"] = np.array([0.9, 0.6, 0.2], dtype=np.float32)
            out["vp.pred_noise"] = torch.stack(noises).numpy()
            out["vp.pred_traj"] = torch.stack(traj).detach().numpy()
    path = os.path.join(ROOT, "tests", "golden", "f3_variants.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
