"""Two identically seeded single-GPU trainers stepping in turn (eager, then captured graphs): are their parameters equal?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import disable_dropout
from moleculesde_amd import pretrain, hip
import moleculesde_amd.geom3d as G
from moleculesde_amd.synthetic import make_batch

dev = torch.device("cuda", 0)
args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=1, emb_dim=64)
MODE = sys.argv[1] if len(sys.argv) > 1 else "alt"


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


def trainer(seed):
    torch.manual_seed(seed)
    tr = pretrain.Trainer(args, dev)
    for m in tr.models.values():
        disable_dropout(m)
    tr.noise = FixedNoise(5)
    for k in ("SDE_2Dto3D_model", "SDE_3Dto2D_model"):
        if k in tr.models:
            tr.models[k].noise = tr.noise
    return tr


b = G.prepare_batch(make_batch(24, seed=31), dev)
for trial in range(int(os.environ.get("TRIALS", "6"))):
    a, c = trainer(3), trainer(3)
    for k in a.models:
        c.models[k].load_state_dict(a.models[k].state_dict())
    if MODE in ("grads", "keepall"):
        a.step(b); a.step(b)
        kept = []
        if MODE == "keepall":
            # every tensor the capture allocates through torch.empty / zeros / empty_like stays referenced: the graph's memory
            # pool never hands the same bytes to two tensors, and every intermediate can be read after a replay
            import traceback
            orig = {n: getattr(torch, n) for n in ("empty", "zeros", "empty_like", "zeros_like", "full")}

            def wrap(fn):
                def f(*aa, **kw):
                    t = fn(*aa, **kw)
                    if t.is_cuda:
                        fr = [x for x in traceback.extract_stack()[:-1] if "moleculesde_amd" in x.filename][-3:]
                        kept.append((t, " / ".join("%s:%s:%d" % (os.path.basename(x.filename), x.name, x.lineno) for x in fr)))
                    return t
                return f
            for n, fn in orig.items():
                setattr(torch, n, wrap(fn))
        a.capture(b)
        if MODE == "keepall":
            for n, fn in orig.items():
                setattr(torch, n, fn)
            print("tensors kept alive:", len(kept), flush=True)
        snap = [t.clone() for t in (a.opt.flat_p, a.opt.m, a.opt.v, a.opt.step_dev, a.step_counter)]
        names = [(mk + "." + n, p) for mk in a.models for n, p in a.models[mk].named_parameters()]
        seen = {}
        for it in range(12):
            with torch.no_grad():
                for t, s0 in zip((a.opt.flat_p, a.opt.m, a.opt.v, a.opt.step_dev, a.step_counter), snap):
                    t.copy_(s0)
            hip.invalidate_weight_copies()
            a.step_graph(b)
            torch.cuda.synchronize()
            for n, p in names:
                if p.grad is not None:
                    seen.setdefault(n, []).append(p.grad.clone())
            if kept:
                sums = torch.stack([t.view(-1).view(torch.int32).to(torch.int64).sum() if t.dtype == torch.float32 and t.numel() else
                                    torch.zeros((), dtype=torch.int64, device=dev) for t, _ in kept])
                seen.setdefault("__kept__", []).append(sums)
                watch = [j for j, (t, site) in enumerate(kept) if any(w in site for w in ("node_forward:231", "edge_forward:129", "forward:308", "node_forward:230", "edge_forward:128"))]
                seen.setdefault("__watch__", []).append({j: kept[j][0].clone() for j in watch})
        if kept:
            ks = seen.pop("__kept__")
            ws = seen.pop("__watch__")
            for r in range(2, len(ws)):
                for j in sorted(ws[r]):
                    x, y = ws[r][j], ws[r - 1][j]
                    if not torch.equal(x, y) and x.dim() == 2:
                        dd = (x != y) & ~(torch.isnan(x) & torch.isnan(y))
                        nz = dd.nonzero()
                        if nz.numel():
                            rows = sorted(set(nz[:, 0].tolist())); cols = sorted(set(nz[:, 1].tolist()))
                            print("      replay %d vs %d  #%d %s: %d elements differ, %d rows (first %s), cols %s, max |d| %.3g"
                                  % (r, r - 1, j, kept[j][1].split(" / ")[-2], nz.size(0), len(rows), rows[:6], cols[:40],
                                     float((x - y)[dd].abs().max())), flush=True)
            first = None
            for r in range(2, len(ks)):
                d = (ks[r] != ks[r - 1]).nonzero().view(-1)
                if d.numel():
                    print("   replay", r, "vs", r - 1, ": intermediates that differ:", d.numel(), flush=True)
                    for j in d[:10].tolist():
                        print("        #%d %s %s" % (j, kept[j][1], tuple(kept[j][0].shape)), flush=True)
        var = []
        for n, gs in seen.items():
            k = sum(1 for g in gs[1:] if not torch.equal(g, gs[0]))
            if k:
                var.append((n, k, max(float((g - gs[0]).abs().max()) for g in gs[1:]), float(gs[0].abs().max())))
        print("grads trial", trial, "parameters whose gradient varies over 12 replays from the same state:", len(var), "of", len(seen), flush=True)
        for v in var[:400]:
            print("    %-70s differs in %2d replays, max |dg| %.3g (|g| max %.3g)" % v, flush=True)
        continue
    if MODE == "solo":
        a.step(b); c.step(b)
        a.capture(b)
        def check_planes(tag):
            nbad = 0
            for key, ent in hip._WT.items():
                if key[0] != "bf16x3":
                    continue
                w = ent["refs"][0]()
                rows, cols = key[3], key[4]
                f = (ent["wt"].to(torch.int32) << 16).view(torch.float32).sum(0)[:, :cols]
                want = w.detach().t() if key[1] else w.detach()
                if not torch.equal(f, want):
                    nbad += 1
                    print("   ", tag, "planes != weight:", key[1], rows, cols, "elements off", int((f != want).sum()), flush=True)
            return nbad
        for it in range(3):
            a.step_graph(b)
            torch.cuda.synchronize()
            print("  after replay", it, "entries with wrong planes:", check_planes("replay %d" % it), flush=True)
        for _ in range(3):
            c.step(b)
        torch.cuda.synchronize()
        print("solo trial", trial, "graph alone vs eager afterwards %.3e" % float((a.opt.flat_p - c.opt.flat_p).norm() / c.opt.flat_p.norm()), flush=True)
        continue
    if MODE == "three":
        e = trainer(3)
        for k in a.models:
            e.models[k].load_state_dict(a.models[k].state_dict())
        a.step(b); c.step(b); e.step(b)
        a.capture(b); c.capture(b)
        def diff_names(x, y):
            out = []
            for mk in x.models:
                for (n, p), (_, q) in zip(x.models[mk].named_parameters(), y.models[mk].named_parameters()):
                    dd = float((p - q).abs().max())
                    if dd > 0:
                        out.append((mk + "." + n, dd, int(((p - q).abs() > 0).sum()), p.numel()))
            return out
        for it in range(3):
            a.step_graph(b); c.step_graph(b); e.step(b)
            torch.cuda.synchronize()
            for tag, x in (("first", a), ("second", c)):
                dn = diff_names(x, e)
                if dn:
                    print("  replay", it, tag, "captured graph differs from eager in", len(dn), "parameters; first:", dn[:6], flush=True)
        rel = lambda x, y: float((x.opt.flat_p - y.opt.flat_p).norm() / y.opt.flat_p.norm())
        print("three trial", trial, "first-captured vs eager %.3e   second-captured vs eager %.3e" % (rel(a, e), rel(c, e)), flush=True)
        continue
    if MODE == "onestream":
        a.overlap_streams = c.overlap_streams = False
    a.step(b); c.step(b)
    a.capture(b)
    if MODE != "eagerref":
        c.capture(b)
    for _ in range(3):
        a.step_graph(b)
        if MODE == "sync":
            torch.cuda.synchronize()
        if MODE == "eagerref":
            c.step(b)
        else:
            c.step_graph(b)
        if MODE == "sync":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    d = float((a.opt.flat_p - c.opt.flat_p).norm() / c.opt.flat_p.norm())
    print(MODE, "trial", trial, "distance %.3e" % d, flush=True)
