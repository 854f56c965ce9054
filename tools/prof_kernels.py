"""Launch the two roofline kernels of bench.py (CFConv aggregate, fused CFConv forward) on the bs-256
synthetic batch, 20 times each, for rocprofv3 (--kernel-trace --stats, or --pmc FETCH_SIZE / WRITE_SIZE
in separate passes as MI355X_MICROARCH.md prescribes)."""
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
with torch.no_grad():
    rplan, dist = hip.radius_plan(b.positions, pl.batch_i32, pl.mol_ptr, sch.cutoff, pl.E_r_cap, 32)
    N = b.x.size(0)
    x1 = torch.randn(N, 128, device=dev); Wf = torch.randn(rplan.E, 128, device=dev); C = torch.rand(rplan.E, device=dev)
    for _ in range(20):
        hip.cfconv_aggregate(x1, Wf, C, rplan)
    torch.cuda.synchronize()
    for _ in range(20):
        hip.cfconv_fused_forward(x1, dist, rplan, blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight, blk.mlp[2].bias,
                                 de.offset, de.coeff, sch.cutoff)
    torch.cuda.synchronize()
print("E", int(rplan.rowptr[-1]), "N", N)
