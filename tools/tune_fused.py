"""Time the fused CFConv forward for several chunks_per_wg values, and the backward kernels (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip, plan as P, pretrain
from moleculesde_amd.geom3d import prepare_batch
from moleculesde_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
torch.manual_seed(0)
args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=0)
tr = pretrain.Trainer(args, dev)
b = prepare_batch(make_batch(256, seed=0), dev)
sch = tr.models["model_3D"]; pl = P.get_plan(b); blk = sch.interactions[0]; de = sch.distance_expansion
def t(fn, it=50):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it * 1e3
with torch.no_grad():
    rplan, dist = hip.radius_plan(b.positions, pl.batch_i32, pl.mol_ptr, sch.cutoff, pl.E_r_cap, 32)
    N = b.x.size(0)
    x1 = torch.randn(N, 128, device=dev)
    for cpw in (0, 1, 2, 3, 4, 6, 8):
        for wf in (False,):
            us = t(lambda: hip.cfconv_fused_forward(x1, dist, rplan, blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight,
                                                    blk.mlp[2].bias, de.offset, de.coeff, sch.cutoff, chunks_per_wg=cpw, want_filter=wf))
            print(f"fused fwd cpw={cpw} want_filter={wf}: {us:.1f} us")
