"""Host time of one hipGraph replay of the pretrain step (how long hipGraphLaunch holds the launching thread) against the
GPU time of the step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from moleculesde_amd import pretrain
from moleculesde_amd.geom3d import prepare_batch
from moleculesde_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
torch.manual_seed(0)
tr = pretrain.Trainer(pretrain.readme_args(SDE_coeff_generative_3Dto2D=0), dev)
b = prepare_batch(make_batch(256, seed=0), dev)
for _ in range(3):
    tr.step(b)
tr.capture(b)
for _ in range(5):
    tr.step_graph(b)
torch.cuda.synchronize()
hs = []
t_all = time.perf_counter()
for _ in range(50):
    t0 = time.perf_counter()
    tr.step_graph(b)
    hs.append(time.perf_counter() - t0)
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
hs.sort()
print(f"host time per replay: median {hs[25] * 1e3:.3f} ms  min {hs[0] * 1e3:.3f}  max {hs[-1] * 1e3:.3f};  step {t_all / 50 * 1e3:.3f} ms")
# a replay issued to an idle GPU: launch-to-finish
lat = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step_graph(b)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    lat.append((t1 - t0, time.perf_counter() - t0))
print("idle-GPU replay: host return %.3f ms, finished %.3f ms" % (sorted(x[0] for x in lat)[5] * 1e3, sorted(x[1] for x in lat)[5] * 1e3))
