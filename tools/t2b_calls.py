import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moleculesde_amd import pretrain
import moleculesde_amd.geom3d as G
from moleculesde_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=1, emb_dim=64)
torch.manual_seed(3)
tr = pretrain.Trainer(args, dev)
b = G.prepare_batch(make_batch(24, seed=31), dev)
tr.step(b)
os.environ["MSDE_DEBUG_T2B"] = "1"
print("==== second step", flush=True)
tr.step(b)
