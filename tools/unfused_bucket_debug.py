import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import moleculesde_amd.geom3d as G
from moleculesde_amd.geom3d import gnn as GN
from moleculesde_amd import pretrain, bucket as BK, hip
from moleculesde_amd.synthetic import make_batch
from helpers import disable_dropout
import test_gpu_plan as TP
GN.FUSE_GIN_LAYER = sys.argv[1] == "fused"
dev = torch.device("cuda", 0)
args = pretrain.readme_args(emb_dim=64, SDE_coeff_generative_3Dto2D=0)
torch.manual_seed(21)
tr = pretrain.Trainer(args, dev)
tr.overlap_streams = False
for m in tr.models.values():
    disable_dropout(m)
cpu_b = make_batch(48, seed=23)
need = BK.raw_sizes(cpu_b)
N = need["N"]
perm = torch.randperm(N, generator=torch.Generator().manual_seed(1))


def run(batch, n_rows):
    noise = TP._fixed_noise(G, dev)
    p1 = torch.arange(n_rows); p1[:N] = perm
    p2 = torch.arange(n_rows); p2[:N] = perm.flip(0)
    noise.randperm_pair = lambda n, device: (p1.to(device).int(), p2.to(device).int())
    tr.noise = noise
    tr.models["SDE_2Dto3D_model"].noise = noise
    tr.opt.zero_grad()
    loss, parts = tr.losses(batch)
    tr._backward(loss)
    torch.cuda.synchronize()
    return tr.opt.gather_grads().clone()


sd0 = {k: {n: v.clone() for n, v in m.state_dict().items()} for k, m in tr.models.items()}
g_e = run(G.prepare_batch(cpu_b.clone(), dev), N)
for k, m in tr.models.items():
    m.load_state_dict(sd0[k])
g_e2 = run(G.prepare_batch(cpu_b.clone(), dev), N)
print("exact vs exact again: max rel", float((g_e - g_e2).norm() / g_e.norm()))
GN.FUSE_GIN_LAYER = not GN.FUSE_GIN_LAYER
for k, m in tr.models.items():
    m.load_state_dict(sd0[k])
g_o = run(G.prepare_batch(cpu_b.clone(), dev), N)
GN.FUSE_GIN_LAYER = not GN.FUSE_GIN_LAYER
names0 = {id(p): f"{k}.{n}" for k, m in tr.models.items() for n, p in m.named_parameters()}
nm0 = max(float(g_e[o:o + sz].norm()) for o, sz in zip(tr.opt.offsets, tr.opt.sizes))
w0 = sorted(((float((g_o[o:o + sz].double() - g_e[o:o + sz].double()).norm()) / max(float(g_e[o:o + sz].double().norm()), 1e-3 * nm0), names0[id(p)])
             for p, o, sz in zip(tr.opt.params, tr.opt.offsets, tr.opt.sizes)), reverse=True)
print("exact: this mode vs the other mode:", [(round(w, 5), n.replace("model_2D.", "")) for w, n in w0[:4]])
names = {id(p): f"{k}.{n}" for k, m in tr.models.items() for n, p in m.named_parameters()}
nmax = max(float(g_e[o:o + sz].norm()) for o, sz in zip(tr.opt.offsets, tr.opt.sizes))
for label, pads in (("none", (0, 0, 0, 0, 0)), ("N only", (300, 0, 0, 0, 0)), ("E_b only", (0, 600, 0, 0, 0)), ("E_e only", (0, 0, 3000, 0, 0)),
                    ("all", (300, 600, 3000, 2500, 3000))):
    for k, m in tr.models.items():
        m.load_state_dict(sd0[k])
    caps = BK.Caps(need["B"], N + pads[0], need["E_b"] + pads[1], need["E_e"] + pads[2], need["E_r"] + pads[3], need["P"] + pads[4], 24)
    bk = BK.Bucket(caps, dev)
    bk.load(BK.pack_raw(cpu_b, caps))
    bk.build_plan_on_device()
    with bk.bounds():
        g_b = run(bk.batch, caps.N)
    worst = []
    for p, o, sz in zip(tr.opt.params, tr.opt.offsets, tr.opt.sizes):
        a, b = g_b[o:o + sz].double(), g_e[o:o + sz].double()
        worst.append((float((a - b).norm()) / max(float(b.norm()), 1e-3 * nmax), names[id(p)]))
    worst.sort(reverse=True)
    print(sys.argv[1], label, [(round(w, 5), n.replace("model_2D.", "")) for w, n in worst[:3]])
    hip.clear_row_bounds()
