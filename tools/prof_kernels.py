"""Launch the roofline kernels of bench.py on the bs-256 synthetic batch (seed 0), 20 times each, for rocprofv3:
`--kernel-trace --stats`, or `--pmc <counters>` in SEPARATE passes as MI355X_MICROARCH.md prescribes (FETCH_SIZE and
WRITE_SIZE cannot share a pass; SQ counters 8 per pass).  Kernels: the fused CFConv forward in TRAINING mode (filter
rows written), the fused CFConv weight-gradient kernel at the step's width and at full width, the CFConv aggregate
input-gradient kernel (the in-step HBM-bound message-passing kernel), and -- with `head` -- the dense 3D->2D head
kernels on the same batch.  tools/pmc_summary.py turns the counter CSVs into profiles/r02_pmc_*.json."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip, _lib, plan as P, pretrain
from moleculesde_amd.geom3d import prepare_batch
from moleculesde_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
torch.manual_seed(0)
args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=0)
tr = pretrain.Trainer(args, dev)
b = prepare_batch(make_batch(256, seed=0), dev)
sch = tr.models["model_3D"]; pl = P.get_plan(b); blk = sch.interactions[0]; de = sch.distance_expansion
REP = 20
with torch.no_grad():
    rplan, dist = hip.radius_plan(b.positions, pl.batch_i32, pl.mol_ptr, sch.cutoff, pl.E_r_cap, 32)
    N = b.x.size(0)
    x1 = torch.randn(N, 128, device=dev); Wf = torch.randn(rplan.E, 128, device=dev); g = torch.randn(N, 128, device=dev)
    W1, b1, W2, b2 = blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight, blk.mlp[2].bias
    for _ in range(REP):      # training-mode forward: filter rows out
        hip.cfconv_fused_forward(x1, dist, rplan, W1, b1, W2, b2, de.offset, de.coeff, sch.cutoff, want_filter=True)
    torch.cuda.synchronize()
    p, st = hip._p, hip._stream()
    gW1, gb1, gW2, gb2 = torch.empty_like(W1), torch.empty_like(b1), torch.empty_like(W2), torch.empty_like(b1)
    for mw in (pretrain.SIDE_CFCONV_BWD_WGS, 0):      # step width first, then full width
        ws = hip._cf_workspace(rplan.E, 51, dev, mw)
        for _ in range(REP):
            _lib.call("msde_cfconv_fused_bwd_w", p(g), p(x1), p(dist), p(rplan.rowptr), p(rplan.src), p(rplan.dst), p(W1),
                      p(b1), p(W2), p(de.offset), N, 128, 51, rplan.E, float(de.coeff), float(sch.cutoff), mw, p(gW1), p(gb1),
                      p(gW2), p(gb2), p(ws), st)
        torch.cuda.synchronize()
    out = torch.empty(N, 128, device=dev)
    for _ in range(REP):
        _lib.call("msde_cfconv_aggregate_bwd_x", p(g), p(Wf), p(None), p(rplan.rowptr_s), p(rplan.perm_s), p(rplan.dst), N,
                  128, p(out), st)
    torch.cuda.synchronize()
print("E", int(rplan.rowptr[-1]), "N", N, "bwd_w widths", pretrain.SIDE_CFCONV_BWD_WGS, "then full")
if len(sys.argv) > 1 and sys.argv[1] == "head":
    args = pretrain.readme_args()
    tr = pretrain.Trainer(args, dev)
    head = tr.models["SDE_3Dto2D_model"]
    h3 = torch.randn(N, 300, device=dev, requires_grad=True)
    for _ in range(5):
        lx, la = head(h3, b, continuous=True, train=True, reduce_mean=True, anneal_power=0)
        tr.opt.zero_grad()
        (lx + la).backward()
    torch.cuda.synchronize()
    print("head ok", float(lx), float(la))
