"""Does any kernel of the step read LDS it never wrote?  One eager single-stream step is run twice from the same state: once as
is, once with the LDS of every CU filled with a pattern (NaN by default) in front of the library calls whose index lies in
[POISON_FROM, POISON_TO).  Gradients that differ name a kernel that read leftovers.  usage: lds_poison_step.py [from] [to]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import disable_dropout
from moleculesde_amd import pretrain, hip, _lib
import moleculesde_amd.geom3d as G
from moleculesde_amd.synthetic import make_batch
from moleculesde_amd import wcache  # noqa: E402

dev = torch.device("cuda", 0)
tool = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "liblds_canary.so"))
tool.lds_poison_launch.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device=dev)
PATTERN = int(os.environ.get("POISON_PATTERN", "0x7FC00000"), 16)
STATE = {"on": False, "idx": 0, "lo": 0, "hi": 1 << 30, "names": []}


class Proxy:
    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)

    def __getattr__(self, name):
        fn = getattr(object.__getattribute__(self, "_lib"), name)
        if not name.startswith("msde_") or name.endswith(("_supported", "_geometry", "_bytes", "_slabs", "num_cus")):
            return fn

        def wrapped(*a):
            if STATE["on"]:
                i = STATE["idx"]; STATE["idx"] += 1
                STATE["names"].append(name)
                if STATE["lo"] <= i < STATE["hi"]:
                    tool.lds_poison_launch(PATTERN, 512, sink.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            return fn(*a)
        return wrapped


_lib.load()
_lib._lib = Proxy(_lib._lib)
class FixedNoise(G.DeviceNoise):
    def __init__(self, seed):
        self.g0, self.cache = seed, {}

    def _get(self, key, make, device):
        if key not in self.cache:
            import zlib
            self.cache[key] = make(torch.Generator().manual_seed(self.g0 * 1000003 + zlib.crc32(repr(key).encode()))).to(device)
        return self.cache[key].clone()

    def randn_like(self, x):
        return self._get(("n", tuple(x.shape)), lambda g: torch.randn(x.shape, generator=g), x.device)

    def randint(self, high, size, device):
        return self._get(("i", high, tuple(size)), lambda g: torch.randint(0, high, size, generator=g), device)

    def randperm(self, n, device):
        return self._get(("p", n), lambda g: torch.randperm(n, generator=g), device)

    def rand(self, n, device):
        return self._get(("r", n), lambda g: torch.rand(n, generator=g), device)


args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=1, emb_dim=int(os.environ.get("EMB", "64")))
torch.manual_seed(3)
tr = pretrain.Trainer(args, dev)
tr.overlap_streams = False
for m in tr.models.values():
    disable_dropout(m)
tr.noise = FixedNoise(5)
for k in ("SDE_2Dto3D_model", "SDE_3Dto2D_model"):
    if k in tr.models:
        tr.models[k].noise = tr.noise
b = G.prepare_batch(make_batch(int(os.environ.get("MOLS", "24")), seed=31), dev)
tr.step(b); tr.step(b)
snap = [t.clone() for t in (tr.opt.flat_p, tr.opt.m, tr.opt.v, tr.opt.step_dev, tr.step_counter)]
names = [(mk + "." + n, p) for mk in tr.models for n, p in tr.models[mk].named_parameters()]


def run(lo, hi, on):
    with torch.no_grad():
        for t, s0 in zip((tr.opt.flat_p, tr.opt.m, tr.opt.v, tr.opt.step_dev, tr.step_counter), snap):
            t.copy_(s0)
    wcache.invalidate_weight_copies()
    torch.manual_seed(11)
    STATE.update(on=on, idx=0, lo=lo, hi=hi, names=[])
    loss, _ = tr.step(b)
    STATE["on"] = False
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.clone() for n, p in names if p.grad is not None}, list(STATE["names"])


l0, g0, _ = run(0, 0, False)
l1, g1, _ = run(0, 0, False)
print("two plain runs equal:", all(torch.equal(g0[k], g1[k]) for k in g0), l0, l1)
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 30


def differs(lo, hi):
    l2, g2, calls = run(lo, hi, True)
    bad = [k for k in g0 if not torch.equal(g0[k], g2[k])]
    return bad, calls, l2


bad, calls, l2 = differs(lo, hi)
print("library calls in a step:", len(calls), "| poisoned [%d, %d): parameters whose gradient changed: %d" % (lo, min(hi, len(calls)), len(bad)),
      "loss", l2, "vs", l0, flush=True)
if bad and len(sys.argv) <= 1:
    # bisect to single calls: smallest index range that still changes something
    a, z = 0, len(calls)
    found = []
    def search(a, z):
        if z - a == 1:
            found.append(a); return
        mid = (a + z) // 2
        if differs(a, mid)[0]: search(a, mid)
        if differs(mid, z)[0]: search(mid, z)
    search(a, z)
    for i in found[:20]:
        print("   a poisoned LDS in front of call %d (%s) changes the result" % (i, calls[i]), flush=True)
