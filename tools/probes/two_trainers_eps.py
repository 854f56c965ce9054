"""Repro: gradient of the GIN eps parameters from (a) trainer A's captured step, (b) trainer A launched from the host,
(c) a second trainer B with the same parameters, all on the same batch object."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import moleculesde_amd.geom3d as G
from moleculesde_amd import pretrain
from moleculesde_amd.synthetic import make_batch
from helpers import disable_dropout
dev = torch.device("cuda", 0)
args = pretrain.readme_args(emb_dim=64, SDE_coeff_generative_3Dto2D=0)
exec(open(os.path.join(ROOT, "tests", "dp_gpu_worker.py")).read().split("class FixedNoise")[1].split("def trainer")[0].join(["class FixedNoise", ""]))


def trainer(seed):
    torch.manual_seed(seed)
    tr = pretrain.Trainer(args, dev)
    for m in tr.models.values():
        disable_dropout(m)
    tr.noise = FixedNoise(5)
    tr.models["SDE_2Dto3D_model"].noise = tr.noise
    return tr


b = G.prepare_batch(make_batch(24, seed=31), dev)
A, B = trainer(4), trainer(4)
A.adam_outside_graph = True
for k in A.models:
    B.models[k].load_state_dict(A.models[k].state_dict())
A.step(b)
A.capture(b)
for k in A.models:
    B.models[k].load_state_dict(A.models[k].state_dict())
names = {id(p): f"{k}.{n}" for k, m in A.models.items() for n, p in m.named_parameters()}
eps_idx = [(o, names[id(p)]) for p, o, sz in zip(A.opt.params, A.opt.offsets, A.opt.sizes) if names[id(p)].endswith(".eps")]


def eager(tr):
    loss, _ = tr.losses(b)
    tr.opt.zero_grad()
    tr._backward(loss)
    return tr.opt.gather_grads().clone()


for it in range(2):
    g, _, _ = A._graphs[id(b)]
    g.replay(); torch.cuda.synchronize()
    ga = A.opt.flat_g.clone()
    gb = eager(B); torch.cuda.synchronize()
    ga2 = eager(A); torch.cuda.synchronize()
    for o, n in eps_idx:
        print(it, n, "A graph %.6e  A eager %.6e  B eager %.6e" % (float(ga[o]), float(ga2[o]), float(gb[o])))
    print(it, "max |A graph - A eager|", float((ga - ga2).abs().max()), " max |A eager - B eager|", float((ga2 - gb).abs().max()))
