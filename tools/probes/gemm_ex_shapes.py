"""Shapes of the gemm_ex launches of one --full pretrain step (dense 3D->2D head), in launch order."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from moleculesde_amd import pretrain, hip
from moleculesde_amd.geom3d import prepare_batch
from moleculesde_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
tr = pretrain.Trainer(pretrain.readme_args(), dev)
b = prepare_batch(make_batch(256, seed=0), dev)
tr.step(b)
log = []
orig = hip.gemm_ex
def spy(A, B, out, **kw):
    N = kw.get("N") or (B.size(1) if kw.get("b_kmajor") else B.size(0))
    K = kw.get("K") or A.size(1)
    log.append((A.size(0), int(N), int(K), int(kw.get("K2") or (kw["A2"].size(1) if kw.get("A2") is not None else 0)),
                kw.get("groups", 1), bool(kw.get("b_kmajor")), kw.get("act"), kw.get("dact_from") is not None))
    return orig(A, B, out, **kw)
hip.gemm_ex = spy
import moleculesde_amd.geom3d.dense_head as dh
if hasattr(dh, "hip"):
    dh.hip.gemm_ex = spy
tr.step(b)
torch.cuda.synchronize()
c = collections.Counter(log)
print(len(log), "gemm_ex launches")
for k, v in sorted(c.items(), key=lambda kv: -kv[0][0] * kv[0][1] * (kv[0][2] + kv[0][3]) * kv[0][4] * kv[1]):
    print("x%d  M=%6d N=%4d K=%4d K2=%4d groups=%d kmajor=%d act=%s dact=%d" % ((v,) + k))
