"""Launch the roofline kernels of bench.py (round 4) on the bs-256 synthetic batch (seed 0), 20 times each, for rocprofv3
`--pmc <counters>` in SEPARATE passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE cannot share a pass): the 2-D tiled
GEMM (gemm_t2) as the GIN layer's second product runs it (BatchNorm + ReLU on the A fragments, statistics in the epilogue, 3588 x 300 x 600), the
pair CFConv filter kernel, the pair aggregation and the pair form of the weight-gradient kernel.  tools/pmc_summary_r04.py
turns the counter CSVs into profiles/r04_pmc_counters.json."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip, _lib, plan as P, pretrain
from moleculesde_amd.geom3d import prepare_batch
from moleculesde_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
torch.manual_seed(0)
tr = pretrain.Trainer(pretrain.readme_args(SDE_coeff_generative_3Dto2D=0), dev)
b = prepare_batch(make_batch(256, seed=0), dev)
sch = tr.models["model_3D"]; pl = P.get_plan(b); blk = sch.interactions[0]; de = sch.distance_expansion
REP = 20
with torch.no_grad():
    N = b.x.size(0)
    D, H = 300, 600
    z1 = torch.randn(N, H, device=dev); Wnk = torch.randn(D, H, device=dev) / H ** 0.5
    sc, sh, b2 = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev) * 0.1, torch.randn(D, device=dev)
    a1, z2 = torch.empty(N, H, device=dev), torch.empty(N, D, device=dev)
    strips, _ = hip.rs_geometry(N, D, H)
    stt = torch.empty(strips, 2, D, device=dev)
    for _ in range(REP):
        hip.gemm_node(z1, Wnk, z2, True, D, H, bias=b2, axf="affine", xf=(sc, sh), relu=True, A_out=a1, stats=stt,
                      stats_mode="bnfwd")
    torch.cuda.synchronize()
    pp = hip.pair_plan(b.positions, pl, sch.cutoff)
    x1 = torch.randn(N, 128, device=dev); g = torch.randn(N, 128, device=dev)
    W1, b1, W2, b2f = blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight, blk.mlp[2].bias
    for _ in range(REP):
        agg, Wf = hip.cfconv_pair_forward(x1, pp, W1, b1, W2, b2f, de.offset, de.coeff, sch.cutoff)
    torch.cuda.synchronize()
    p, st = hip._p, hip._stream()
    ws = hip._cf_workspace(pp.P, 51, dev, 0)
    for _ in range(REP):
        _lib.call("msde_cfconv_pair_bwd_w", p(g), p(x1), p(pp.pd), p(pp.count), p(pp.pi), p(pp.pj), p(W1), p(b1), p(W2),
                  p(de.offset), N, 128, 51, pp.P, float(de.coeff), float(sch.cutoff), 0, p(None), p(None), p(None), p(None),
                  p(ws), st)
    torch.cuda.synchronize()
print("N", N, "pairs", int(pp.count[0]))
