"""What each part of the two-stream pretrain step costs: the captured per-shape step (bs 256, resident batch, no plan build)
with (a) everything (configs[1]), (b) the contrastive term off (no SchNet: only the main chain GIN -> 2D->3D), (c) the 2D->3D
term off (GIN + SchNet + contrastive), (d) everything on ONE stream."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import moleculesde_amd.geom3d as G
from moleculesde_amd import pretrain
from moleculesde_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
b = G.prepare_batch(make_batch(256, seed=17), dev)


def run(name, overlap=True, **kw):
    torch.manual_seed(0)
    tr = pretrain.Trainer(pretrain.readme_args(SDE_coeff_generative_3Dto2D=0, **kw), dev)
    tr.overlap_streams = overlap
    for _ in range(3):
        tr.step(b)
    tr.capture(b)
    for _ in range(5):
        tr.step_graph(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        tr.step_graph(b)
    torch.cuda.synchronize()
    print(f"{name:60s} {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step", flush=True)


run("configs[1], two streams")
run("contrastive term off (GIN + 2D->3D: the main chain alone)", SDE_coeff_contrastive=0)
run("2D->3D term off (GIN + SchNet + contrastive)", SDE_coeff_generative_2Dto3D=0)
run("configs[1], one stream", overlap=False)
