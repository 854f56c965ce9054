"""Undistorted timeline of one hipGraph replay of the pretrain step: device timestamps (msde_debug_stamp, 100 MHz) captured
into the graph at the phase boundaries of both streams.  Prints microseconds from the start of the replay, median over
replays.  usage: python tools/probes/step_timeline.py [--bucket] [--full] [--no_cl]"""
import os, sys, torch, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from moleculesde_amd import pretrain, hip, bucket as BK
from moleculesde_amd.geom3d import prepare_batch
from moleculesde_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
torch.manual_seed(0)
full = "--full" in sys.argv
kw = dict(SDE_coeff_contrastive=0) if "--no_cl" in sys.argv else {}      # --no_cl: the main chain alone (no SchNet, no contrastive term)
tr = pretrain.Trainer(pretrain.readme_args(**kw) if full else pretrain.readme_args(SDE_coeff_generative_3Dto2D=0, **kw), dev)
hip.enable_stamps(dev)
cpu = [make_batch(256, seed=s) for s in range(4)]
if "--bucket" in sys.argv:
    caps = BK.Caps.covering([BK.raw_sizes(b) for b in cpu])
    bk = tr.make_bucket(caps)
    blobs = [BK.pack_raw(b, caps).to(dev) for b in cpu]
    tr.capture_bucket(bk, blobs[0])
    run = lambda i: tr.step_bucket(bk, blobs[i % 4])
else:
    b = prepare_batch(cpu[0], dev)
    for _ in range(3):
        tr.step(b)
    tr.capture(b)
    run = lambda i: tr.step_graph(b)
for i in range(10):
    run(i)
torch.cuda.synchronize()
rows = []
import time
t0 = time.perf_counter()
for i in range(30):
    run(i)
    torch.cuda.synchronize()
    rows.append(hip.read_stamps())
print("ms/step (synchronised each step): %.3f" % ((time.perf_counter() - t0) / 30 * 1e3))
names = list(rows[0].keys())
base = "step_start"
med = {n: statistics.median((r[n] - r[base]) / 100.0 for r in rows) for n in names}
for n, v in sorted(med.items(), key=lambda kv: kv[1]):
    print(f"{v:9.1f} us  {n}")
