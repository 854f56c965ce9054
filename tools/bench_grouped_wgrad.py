"""The grouped weight-gradient launch of one pretrain step, alone: the problem list of profiles/r04_wgrad_problems_default.txt
(53 problems, 26 GFLOP; `full`: + the dense head's problems) with operands of their own, timed as a captured hipGraph of REP
launches of msde_linear_bwd_w_grouped, and checked against fp64 products of two of the problems.  MSDE_TOOL_LIB=<.so>: a
variant build (tools/build_variant.sh).

    python tools/bench_grouped_wgrad.py [full] [once]      # `once`: one launch, for rocprofv3 --pmc passes
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moleculesde_amd import _lib, hip

if os.environ.get("MSDE_TOOL_LIB"):           # a variant build (tools/build_variant.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["MSDE_TOOL_LIB"])
dev = torch.device("cuda", 0)
DEFAULT = [(3588, 300, 600, 5), (3588, 600, 300, 5), (3588, 300, 300, 10), (3588, 300, 128, 6), (3588, 128, 300, 6),
           (35186, 128, 64, 2), (35186, 32, 300, 1), (70372, 32, 128, 1), (35186, 128, 32, 1), (35186, 32, 72, 1),
           (35186, 32, 64, 1), (3588, 128, 32, 4), (35186, 32, 32, 1), (3588, 32, 300, 1), (3588, 32, 32, 8)]
# + the dense 3D->2D head (profiles/r05: 157 problems, 35.1 GFLOP): node MLP 364 -> 728 -> 728 -> 119, the edge layers' skinny
# node-level products and the pair-level ones (52 680 pair rows)
FULL_EXTRA = [(3588, 728, 364, 1), (3588, 728, 728, 1), (3588, 119, 728, 1), (3588, 16, 364, 24), (3588, 16, 16, 24),
              (52680, 32, 32, 8), (52680, 16, 32, 8), (3588, 64, 364, 4), (3588, 364, 300, 2), (52680, 60, 30, 2),
              (52680, 1, 60, 1), (3588, 32, 64, 16), (3588, 300, 300, 2)]
probs = list(DEFAULT) + (FULL_EXTRA if "full" in sys.argv else [])
if os.environ.get("MSDE_WG_SKIP"):            # "M,N,K;M,N,K": problems left out (what a class of problems costs the launch)
    skip = {tuple(int(v) for v in t.split(",")) for t in os.environ["MSDE_WG_SKIP"].split(";")}
    probs = [p_ for p_ in probs if p_[:3] not in skip]
ONCE = "once" in sys.argv
REP = 1 if ONCE else 10

lib = _lib.load()
p = hip._p
ROW = 16
ops, rows = [], []
gflop = nbytes = 0.0
for (M, N, K, cnt) in probs:
    for _ in range(cnt):
        g = torch.randn(M, N, device=dev)
        x = torch.randn(M, K, device=dev)
        sp = int(lib.msde_linear_bwd_w_splits(M, N, K))
        slab = torch.empty(sp * (N * K + N), device=dev)
        ops.append((g, x, slab, M, N, K, sp))
        gflop += 2.0 * M * N * K / 1e9
        nbytes += 4.0 * M * (N + K)
n = len(ops)
host_prob = torch.zeros(n, ROW, dtype=torch.int64)
host_pre = torch.zeros(n + 1, dtype=torch.int32)
ops.sort(key=lambda t: -(t[3] / t[6]))
tot = 0
for r, (g, x, slab, M, N, K, sp) in enumerate(ops):
    nb = lib.msde_linear_bwd_w_describe_ld(p(g), N, p(x), K, M, N, K, 1, p(slab), p(None), ctypes.c_void_p(host_prob[r].data_ptr()))
    assert nb > 0
    host_pre[r] = tot
    tot += nb
host_pre[n] = tot
dprob, dpre = host_prob.to(dev), host_pre.to(dev)
st = hip._stream


def natural():
    _lib.call("msde_linear_bwd_w_grouped_ex", p(dprob), p(dpre), n, tot, 0, st())


def gtime(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP):
                fn()
        g.replay(); s.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(5):
            g.replay()
        b.record(s); b.synchronize()
    return a.elapsed_time(b) / (5 * REP) * 1e3


print("problems %d  tiles %d  %.2f GFLOP  operands %.1f MB  slabs %.1f MB" %
      (n, tot, gflop, nbytes / 1e6, sum(t[2].numel() for t in ops) * 4 / 1e6))
if ONCE:
    natural(); torch.cuda.synchronize()
    sys.exit(0)
for t in ops:
    t[2].fill_(float("nan"))
natural(); torch.cuda.synchronize()
worst = 0.0
seen, picks = set(), []
for t in ops:                       # one problem of every distinct shape (every tile shape of the launch gets checked)
    if t[3:6] not in seen:
        seen.add(t[3:6])
        picks.append(t)
for (g, x, slab, M, N, K, sp) in picks:
    gw = slab[:sp * N * K].view(sp, N, K).double().sum(0)
    gb = slab[sp * N * K:].view(sp, N).double().sum(0)
    ref, refb = g.double().t() @ x.double(), g.double().sum(0)
    e = max(float((gw - ref).abs().max() / ref.abs().max()), float((gb - refb).abs().max() / refb.abs().max()))
    worst = max(worst, e)
    print("check %dx%dx%d (%d splits): max err / scale vs fp64 %.2e" % (M, N, K, sp, e))
assert worst < 1e-4 and all(torch.isfinite(t[2]).all() for t in ops), worst
for rep in range(3):
    us = gtime(natural)
    print("grouped launch %7.1f us  %6.1f TFLOP/s  %.3f of 157.3" % (us, gflop / us * 1e3, gflop / us * 1e3 / 157.3))
if "probe" in sys.argv:          # a -DLG_PROBE build: cycles of wave 0 per interior K tile, by phase (average over all workgroups)
    buf = (ctypes.c_ulonglong * 8)()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    raw.msde_debug_lg_probe(None, 1)
    natural(); torch.cuda.synchronize()
    raw.msde_debug_lg_probe(buf, 0)
    v = [int(x) for x in buf]
    tiles = max(v[5], 1)
    names = ("barrier 1 (previous tile consumed)", "global-load wait + LDS stores issued", "barrier 2 (stores landed)",
             "next loads + 32 LDS reads issued", "MFMA phase (first operand wait + 16 MFMAs issued)")
    tot_c = sum(v[:5])
    print("K tiles probed %d; cycles per K tile (s_memtime ticks = shader cycles): total %.0f (16 MFMAs = 1024)" % (tiles, tot_c / tiles))
    for nm, c in zip(names, v[:5]):
        print("  %-52s %7.0f  %4.1f %%" % (nm, c / tiles, 100.0 * c / tot_c))
