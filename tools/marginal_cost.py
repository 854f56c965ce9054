"""Marginal cost of the parts of the pretrain step under hipGraph replay (GPU box): time the full step and the
step with one part removed.  Tells which branch is on the critical path of the multi-stream graph."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import pretrain
from moleculesde_amd.geom3d import prepare_batch
from moleculesde_amd.synthetic import make_batch
dev = torch.device("cuda", 0)

def run(tag, overlap=True, skip_schnet=False, geo_stream=True, **over):
    torch.manual_seed(0)
    args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=0, **over)
    tr = pretrain.Trainer(args, dev)
    tr.overlap_streams = overlap
    if not overlap or not geo_stream:
        tr.models["SDE_2Dto3D_model"].side_stream = None
    if skip_schnet:
        sch = tr.models["model_3D"]
        N = 3588
        fake = torch.zeros(N, 300, device=dev)
        sch.forward = lambda *a, **k: (None, fake)
    b = prepare_batch(make_batch(256, seed=0), dev)
    for _ in range(3):
        tr.step(b)
    tr.capture(b)
    for _ in range(5):
        tr.step_graph(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        tr.step_graph(b)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{tag:40s} {ms:7.3f} ms/step", flush=True)
    return ms

full = run("full (GIN + SchNet + CL + 2D->3D)")
if "--streams" in sys.argv:
    run("SchNet stream only (no geometry stream)", geo_stream=False)
    run("full again")
    run("SchNet stream only again", geo_stream=False)
    run("single stream", overlap=False)
    sys.exit(0)
no23 = run("without 2D->3D", SDE_coeff_generative_2Dto3D=0)
nocl = run("without CL (SchNet forward only)", SDE_coeff_contrastive=0)
nosch = run("without CL and without SchNet", SDE_coeff_contrastive=0, skip_schnet=True)
ser = run("full, single stream", overlap=False)
