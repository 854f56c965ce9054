"""Graph-timed launches of the fused CFConv weight-gradient kernel (csrc/cfconv_fused_bwd.hip) on the bs-256 batch:
full width and the step's width.  MSDE_CFBWD_DBG=<mask> removes phases (1 gathers, 2 rbf, 4 ssp/sigmoid, 8 gW2 MFMAs,
16 g_h1 MFMAs, 32 gW1 MFMAs, 64 pre1 MFMAs, 128 slab write) to find what the time is made of."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip, _lib, plan as P, pretrain
from moleculesde_amd.geom3d import prepare_batch
from moleculesde_amd.synthetic import make_batch
from tools.bench_gemm_ex import timeit
dev = torch.device("cuda", 0)
torch.manual_seed(0)
b = prepare_batch(make_batch(256, seed=0), dev)
pl = P.get_plan(b)
with torch.no_grad():
    rplan, dist = hip.radius_plan(b.positions, pl.batch_i32, pl.mol_ptr, 10.0, pl.E_r_cap, 32)
    N = b.x.size(0)
    E = int(rplan.rowptr[-1])
    x1 = torch.randn(N, 128, device=dev); g = torch.randn(N, 128, device=dev)
    W1 = torch.randn(128, 51, device=dev) * 0.2; b1 = torch.randn(128, device=dev) * 0.1
    W2 = torch.randn(128, 128, device=dev) * 0.1
    offset = torch.linspace(0, 10, 51, device=dev); coeff = -0.5 / float(offset[1] - offset[0]) ** 2
    p, st = hip._p, hip._stream()
    flop = 2.0 * E * (2 * 128 * 128 + 2 * 128 * 52)
    for mw in (0, 128):
        ws = hip._cf_workspace(rplan.E, 51, dev, mw)
        fn = lambda: _lib.call("msde_cfconv_fused_bwd_w", p(g), p(x1), p(dist), p(rplan.rowptr), p(rplan.src), p(rplan.dst),
                               p(W1), p(b1), p(W2), p(offset), N, 128, 51, rplan.E, coeff, 10.0, mw, p(None), p(None),
                               p(None), p(None), p(ws), hip._stream())
        t = timeit(fn)
        print(f"dbg={os.environ.get('MSDE_CFBWD_DBG', '0'):>4s} pipe={os.environ.get('MSDE_CFBWD_PIPE', '-')} width={mw or 256:4d}  "
              f"{t:7.1f} us  {flop / t / 1e6:6.1f} TF  frac {flop / t / 1e6 / 157.3:.3f}", flush=True)

    # ---- the fused forward (training mode: filter rows written), full width and the step's width
    b2 = torch.randn(128, device=dev) * 0.1
    agg = torch.empty(N, 128, device=dev)
    Wf = torch.empty(rplan.E, 128, device=dev)
    fflop = 2.0 * E * (52 * 128 + 128 * 128)
    for wgs in (None, 256):
        cpw = hip.FUSED_CHUNKS_PER_WG if wgs is None else max(1, -(-((rplan.E + 31) // 32) // int(wgs)))
        fn = lambda: _lib.call("msde_cfconv_fused_fwd", p(x1), p(dist), p(rplan.rowptr), p(rplan.src), p(rplan.dst), p(W1),
                               p(b1), p(W2), p(b2), p(offset), N, 128, 51, rplan.E, coeff, 10.0, cpw, p(agg), p(Wf),
                               hip._stream())
        agg.zero_()
        t = timeit(fn)
        print(f"fwd  width={wgs or 'full':>5}  {t:7.1f} us  {fflop / t / 1e6:6.1f} TF  frac {fflop / t / 1e6 / 157.3:.3f}", flush=True)
