"""Shapes of the weight-gradient problems one step queues for the grouped launch (M = reduction rows, N x K = gradient)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moleculesde_amd import pretrain, _lib
import moleculesde_amd.geom3d as G
from moleculesde_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
FULL = len(sys.argv) > 1 and sys.argv[1] == "full"
log = []


class Proxy:
    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)

    def __getattr__(self, name):
        fn = getattr(object.__getattribute__(self, "_lib"), name)
        if name != "msde_linear_bwd_w_describe_ld":
            return fn

        def wrapped(gY, ldg, X, ldx, M, N, K, *rest):
            log.append((int(M), int(N), int(K), int(ldg), int(ldx)))
            return fn(gY, ldg, X, ldx, M, N, K, *rest)
        return wrapped


_lib.load()
_lib._lib = Proxy(_lib._lib)
args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=1 if FULL else 0)
torch.manual_seed(0)
tr = pretrain.Trainer(args, dev)
b = G.prepare_batch(make_batch(256, seed=0), dev)
tr.step(b)
log.clear()
tr.step(b)
torch.cuda.synchronize()
c = collections.Counter(log)
tot_f = tot_b = 0
print("%8s %5s %5s %5s %5s  count   MFLOP   MB(operands)" % ("M", "N", "K", "ldg", "ldx"))
for (M, N, K, ldg, ldx), n in sorted(c.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[1]):
    f = 2.0 * M * N * K * n / 1e6
    by = 4.0 * M * (N + K) * n / 1e6
    tot_f += f; tot_b += by
    print("%8d %5d %5d %5d %5d  %5d %8.1f %8.1f" % (M, N, K, ldg, ldx, n, f, by))
print("problems", len(log), "total GFLOP %.2f" % (tot_f / 1e3), "operand MB %.1f" % tot_b)
