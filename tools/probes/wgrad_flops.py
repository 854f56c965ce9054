"""FLOPs of the weight-gradient GEMMs of one pretrain step (the grouped launch at the end of the backward pass)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MSDE_WGRAD_OVERLAP"] = "0"
os.environ["MSDE_DEFER_LEAF"] = "0"      # nothing runs beside the grouped launch: its stand-alone duration
os.environ["MSDE_OVERLAP_STREAMS"] = os.environ.get("MSDE_OVERLAP_STREAMS", "1")
from moleculesde_amd import pretrain, hip
from moleculesde_amd.geom3d import prepare_batch
from moleculesde_amd.synthetic import make_batch
from moleculesde_amd import slabs  # noqa: E402
dev = torch.device("cuda", 0)
tr = pretrain.Trainer(pretrain.readme_args(**({} if "--full" in sys.argv else {"SDE_coeff_generative_3Dto2D": 0})), dev)
b = prepare_batch(make_batch(256, seed=0), dev)
tr.step(b)
orig = slabs._SLABS.launch_gemms
def spy(max_wgs=0):
    tot = 0
    rows = []
    for (gY, X, M, N, K, hb, slab) in slabs._SLABS.gemms:
        tot += 2.0 * M * N * K
        rows.append((M, N, K))
    import collections
    c = collections.Counter(rows)
    print("queued GEMMs", len(rows), "GFLOP %.2f" % (tot / 1e9))
    for k, v in sorted(c.items(), key=lambda kv: -2.0 * kv[0][0] * kv[0][1] * kv[0][2] * kv[1])[:60]:
        print("  M=%6d N=%4d K=%4d x%d  %.2f GFLOP" % (k[0], k[1], k[2], v, 2e-9 * k[0] * k[1] * k[2] * v))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(max_wgs)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    print("grouped launch alone: %.1f us -> %.1f TFLOP/s (%.2f of the 157.3 peak)" % (us, tot / us / 1e6, tot / us / 1e6 / 157.3))
    return r
slabs._SLABS.launch_gemms = spy
for _ in range(3):
    tr.step(b)
torch.cuda.synchronize()
