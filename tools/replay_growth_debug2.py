"""The in-suite failure of test_replay_after_workspaces_grew (nopair cross-check) outside pytest: the two tests that have to
run before it, then capture(small) -> step(big) -> replay(small), comparing the GRADIENTS of the first replay with the
all-eager sequence."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import moleculesde_amd.geom3d as G
from moleculesde_amd import hip
from moleculesde_amd.synthetic import make_batch
import test_gpu_models as T
dev = torch.device("cuda", 0)
for a in sys.argv[1:]:
    if "=" in a:
        mod, rest = a.split(":"); attr, val = rest.split("=")
        setattr(__import__(mod, fromlist=["x"]), attr, eval(val))
hip.CFCONV_PAIR = False
if "novariants" not in sys.argv:
    T.test_golden_f3_variants_2d3d(dev, "m01")
    T.test_golden_f3_variants_2d3d(dev, "vp")
if "nofar" not in sys.argv:
    T.test_eager_steps_with_gpu_far_behind_host(dev)


class Fixed(G.DeviceNoise):
    def __init__(self):
        super().__init__(seed=3)
        g = torch.Generator().manual_seed(9)
        self.big, self.ints = torch.randn(4096, 3, generator=g).to(dev), torch.randint(0, 1000, (512,), generator=g).to(dev)
        self.perm = {}

    def randn_like(self, x): return self.big[:x.size(0)].clone()
    def randint(self, high, size, device): return self.ints[:size[0]].clone()

    def randperm_pair(self, n, device):
        if n not in self.perm:
            self.perm[n] = (torch.randperm(n, generator=torch.Generator().manual_seed(n)).int().to(device),
                            torch.randperm(n, generator=torch.Generator().manual_seed(n + 1)).int().to(device))
        return self.perm[n]


from moleculesde_amd import slabs
_retire0 = slabs._retire


def _retire_logged(bufs):
    import traceback
    fr = traceback.extract_stack(limit=3)[0]
    print("   _retire:", [None if b is None else (tuple(b.shape), hex(b.data_ptr())) for b in bufs], "from", fr.name, fr.lineno, "captured", slabs._CAPTURED, flush=True)
    _retire0(bufs)


if "logretire" in sys.argv:
    slabs._retire = _retire_logged        # (hip.py's own callers go through its imported name: patch both)
    slabs._retire = _retire_logged
from moleculesde_amd import pretrain as _pt
_dual0 = _pt.dual_CL
CLSTASH = []


def _dual_logged(X, Y, a, noise, negs=(None, None)):
    CLSTASH.append((X.detach(), Y.detach(), negs[0], negs[1]))           # detached views: the capture's storage stays readable after the replay
    return _dual0(X, Y, a, noise, negs)


if "clstash" in sys.argv:
    _pt.dual_CL = _dual_logged
OPSTASH = []


def _wrap(name):
    f0 = getattr(hip, name)

    def f(*a, **k):
        r = f0(*a, **k)
        outs = r if isinstance(r, (tuple, list)) else (r,)
        for j, o in enumerate(outs):
            if torch.is_tensor(o):
                OPSTASH.append((f"{name}[{j}]", o.detach()))
            elif hasattr(o, "rowptr"):
                OPSTASH.extend([(f"{name}.rowptr", o.rowptr), (f"{name}.src", o.src), (f"{name}.dst", o.dst), (f"{name}.rowptr_s", o.rowptr_s), (f"{name}.perm_s", o.perm_s)])
        return r
    setattr(hip, name, f)


if "opstash" in sys.argv:
    for nm in ("embedding_sum", "radius_plan", "linear_fork", "cfconv_fused", "schnet_tail", "mlp_fused", "segment_reduce"):
        _wrap(nm)
small = G.prepare_batch(make_batch(8, seed=81), dev)
big = G.prepare_batch(make_batch(96, seed=82), dev)
res = []
for use_graph in (True, False):
    tr = T._quiet_trainer(dev, seed=6)
    for m in tr.models.values():
        T.disable_dropout(m)
    tr.noise = Fixed()
    tr.models["SDE_2Dto3D_model"].noise = tr.noise
    names = [f"{mn}.{pn}" for mn, m in tr.models.items() for pn, p in m.named_parameters()]
    params = [p for m in tr.models.values() for p in m.parameters()]
    print("run graph" if use_graph else "run eager", flush=True)
    tr.step(small)
    gstatic = None
    if use_graph:
        print("  capture", flush=True)
        from moleculesde_amd import wcache
        before = set(wcache._WT)
        n_before = len(OPSTASH)
        tr.capture(small)
        OPSTASH_CAP = OPSTASH[n_before:]
        OPN = len(OPSTASH_CAP)
        gstatic = [p.grad for p in params]          # p.grad as the capture left it
        gstatic0 = gstatic
        made = [k for k in wcache._WT if k not in before]
        gone = [k for k in before if k not in wcache._WT]
        print("  weight-copy entries: before capture", len(before), "after", len(wcache._WT), "made inside", len(made), "dropped", len(gone))
        spans = [(e["wt"].data_ptr(), e["wt"].data_ptr() + 4 * e["wt"].numel(), "wt", k[1:3]) for k, e in wcache._WT.items()]
        spans += [(b.data_ptr(), b.data_ptr() + 4 * b.numel(), "keepalive", tuple(b.shape)) for b in slabs._KEEP_ALIVE if hasattr(b, "data_ptr")]
        hits = 0
        for n_, g_ in zip(names, gstatic):
            if g_ is None:
                continue
            a_, b_ = g_.data_ptr(), g_.data_ptr() + 4 * g_.numel()
            for lo, hi, what, info in spans:
                if a_ < hi and lo < b_:
                    hits += 1
                    if hits <= 12:
                        print("   OVERLAP: gradient of", n_, tuple(g_.shape), "with", what, info)
        print("  gradient tensors overlapping a weight copy / parked buffer:", hits)
    if use_graph and "probe" in sys.argv:
        tr.step_graph(small)
        torch.cuda.synchronize()
        s1 = [None if t is None else t.clone() for t in gstatic]
        print("  step(big)", flush=True)
        tr.step(big)
        torch.cuda.synchronize()
        ch = [n_ for n_, a_, b_ in zip(names, s1, gstatic) if a_ is not None and not torch.equal(a_, b_)]
        print("  PROBE: gradient tensors of the graph changed by the EAGER step on the big batch:", len(ch), ch[:6], flush=True)
        sys.exit(0)
    print("  logged sums before step(big):", {k: float(v) for k, v in tr.log.items()}, flush=True)
    print("  step(big)", flush=True)
    tr.step(big)
    print("  logged sums after step(big):", {k: float(v) for k, v in tr.log.items()}, flush=True)
    print("  first small step behind it", flush=True)
    if "dirty" in sys.argv:        # whatever the eager step handed back to the caching allocator now holds NaN
        torch.cuda.synchronize()
        xs = [torch.full((n,), float("nan"), device=dev) for n in (1 << 8, 1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24) for _ in range(24)]
        del xs
        torch.cuda.synchronize()
    if "refresh" in sys.argv and use_graph:
        wcache.refresh_weight_t()
    pre = [None if t is None else t.clone() for t in gstatic] if use_graph else None
    if use_graph and "tables" in sys.argv:
        cs = slabs._SLABS.slots[-1]          # the capture's slot: (host_rows, host_pre, dev_rows, dev_pre, host_prob, host_ppre, dev_prob, dev_ppre)
        torch.cuda.synchronize()
        for nm_, h_, d_ in (("rows", cs[0], cs[2]), ("pre", cs[1], cs[3]), ("prob", cs[4], cs[6]), ("ppre", cs[5], cs[7])):
            eq = torch.equal(h_, d_.cpu())
            nz = int((h_ != 0).any(dim=-1).sum()) if h_.dim() == 2 else int((h_ != 0).sum())
            print(f"  capture slot table {nm_:5s}: device == host image: {eq}; non-zero rows {nz}", flush=True)
        print("  slots:", len(slabs._SLABS.slots), "current slot index", slabs._SLABS.slot_i, "eager_i", slabs._SLABS.eager_i, flush=True)
    if use_graph and "sentinel" in sys.argv:
        print("  arena:", tuple(slabs._SLABS.arena.shape), hex(slabs._SLABS.arena.data_ptr()), "retired", len(slabs._SLABS.retired), flush=True)
        slabs._SLABS.arena.fill_(777.0)
        torch.cuda.synchronize()
    (tr.step_graph if use_graph else tr.step)(small)
    torch.cuda.synchronize()
    if use_graph:
        same = [n_ for n_, a_, b_ in zip(names, pre, gstatic) if a_ is not None and torch.equal(a_.view(torch.int32), b_.view(torch.int32))]
        print("  gradient tensors (p.grad as the capture left it) that the replay did NOT change:", len(same), same[:5], flush=True)
    print("  logged sums after this step:", {k: float(v) for k, v in tr.log.items()}, flush=True)
    if CLSTASH:
        X_, Y_, n1_, n2_ = CLSTASH[1] if use_graph else CLSTASH[-1]      # graph run: the capture's call (2nd); eager: the last
        print("  contrastive inputs of this step: X", tuple(X_.shape), float(X_.double().abs().sum()), "Y", tuple(Y_.shape), float(Y_.double().abs().sum()),
              "neg1", None if n1_ is None else (tuple(n1_.shape), int(n1_.min()), int(n1_.max()), int(n1_.long().sum())),
              "neg2", None if n2_ is None else (tuple(n2_.shape), int(n2_.min()), int(n2_.max()), int(n2_.long().sum())), flush=True)
        CLSTASH.clear()
    if OPSTASH:
        # graph run: the capture's calls are the LAST ones recorded before the eager step on the big batch ran
        print("  operator outputs of this step:", flush=True)
        for nm, t in (OPSTASH_CAP if use_graph else OPSTASH[-OPN:]):
            tt = t.double() if t.is_floating_point() else t.long()
            print(f"     {nm:24s} {str(tuple(t.shape)):14s} abs sum {float(tt.abs().sum()):.6e}", flush=True)
        OPSTASH.clear()
    grads = gstatic if use_graph else [p.grad for p in params]
    res.append(([None if t is None else t.detach().clone() for t in grads], tr.opt.flat_p.clone(), names))
(g0, p0, names), (g1, p1, _) = res
print("parameters after the first small step behind the big one: graph/eager rel diff", float((p0 - p1).norm() / p1.norm()))
bad, off = [], 0
params_ = [pp for m in tr.models.values() for pp in m.parameters()]
order = {id(pp): n for n, pp in zip(names, params_)}
for pp in tr.opt.params:
    n = pp.numel()
    a, b = p0[off:off + n], p1[off:off + n]
    d = float((a - b).norm() / (b.norm() + 1e-30))
    if d > 1e-7:
        bad.append((d, order.get(id(pp), "?"), n))
    off += n
print(len(bad), "of", len(tr.opt.params), "parameters differ after ONE replay; in optimiser order:", [(round(d, 6), n) for d, n, _ in bad][:60])
gb = [(n, float((a - b).norm() / (b.norm() + 1e-30))) for n, a, b in zip(names, g0, g1) if a is not None and b is not None]
print("p.grad as the capture left it vs the eager gradients of the same step:", sum(1 for _, d in gb if d < 1e-5), "equal,",
      sum(1 for _, d in gb if not d < 1e-5), "different (of", len(gb), ")")
shown = 0
for n, a, b in zip(names, g0, g1):
    if a is None or b is None:
        continue
    d = float((a - b).norm() / (b.norm() + 1e-30))
    if not d < 1e-5 and shown < 14:
        shown += 1
        fin = torch.isfinite(a)
        big_ = (a.abs() > 1e6) | ~fin
        print(f"   {n:55s} {str(tuple(a.shape)):12s} entries off by >1e-6: {int(((a - b).abs() > 1e-6 * (b.abs().max() + 1e-30)).sum()):6d} of {a.numel():6d}; huge/non-finite {int(big_.sum()):6d}; "
              f"first bad flat index {int(torch.nonzero(((a - b).abs() > 1e-6 * (b.abs().max() + 1e-30)).flatten())[0]) if d > 0 else -1}; ptr {hex(g0 and a.data_ptr())}")
print("addresses of p.grad (capture) for the first 6 different:", [hex(t.data_ptr()) for t, (n, d) in zip([t for t in gstatic0 if t is not None], gb) if not d < 1e-5][:6])
bad_models = {}
for n, a, b in zip(names, g0, g1):
    if a is None or b is None:
        continue
    d = float((a - b).norm() / (b.norm() + 1e-30))
    key = n.split(".")[0]
    bad_models.setdefault(key, [0, 0])
    bad_models[key][0 if d < 1e-5 else 1] += 1
print("gradients equal / different per model:", bad_models)
k777 = sum(int(((a.abs() > 700) & (a.abs() < 1e6)).sum()) for a in g0 if a is not None)
print("entries of the replay's gradients between 700 and 1e6 (sentinel sums):", k777)
