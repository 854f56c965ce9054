"""bench.py — molecules/s of one MoleculeSDE pretrain step on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): PCQM4Mv2-shaped synthetic batches, 256 molecules per GPU,
GIN(5x300) + SchNet(6 interactions, 128 filters, 51 Gaussians, cutoff 10) + contrastive
(EBM_node_dot_prod, so SchNet receives gradient) + SDEModel2Dto3D_02 VE; one "step" = device-side batch
construction, forward, backward, (all-reduce), Adam, inputs already resident in HBM.  `--full` adds the 3D->2D dense
head (configs[2] per-GPU work).  Weak scaling: every rank owns its own shard of molecules; the only collective is the
all-reduce of the flat gradient (per-model buckets).

Headline (`value`): `--stream` (64) DISTINCT batches streamed through ONE captured hipGraph (capacity bucket, raw
collated arrays in, plans + extended graph built on the device).  Secondary keys under config.stream: the same blobs
from pinned host memory (PCIe-inclusive) and the round-1 mode (4 resident batches, a graph each).

Rank 0 prints ONE JSON line; `roofline` = the kernel with the largest share of the step's GPU time (round 3: the row-strip
fp32-MFMA GEMM on the GIN layer's 3588 x 300 x 600 product), its duration measured INSIDE the captured step with device
timestamps (both streams running), the stand-alone HIP-event timing kept under `standalone`;
`roofline_forward_schnet_sde2d3d` = the north-star forward figure; `config4_sampler` / `config5_md17` = BASELINE.json
configs[3] / configs[4] with the oracle's CPU time beside them; `cpu_baseline` = the oracle port of the pretrain step timed on
the host cores (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: fp32 matrix == fp32 vector peak


def _event_time_ms(fn, iters, stream):
    """Average duration (ms) of fn() over `iters` launches, HIP events on the launch stream."""
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    start.record(stream)
    for _ in range(iters):
        fn()
    end.record(stream)
    end.synchronize()
    return start.elapsed_time(end) / iters


def _pmc_traffic(kernel, rnd="r06"):
    """HBM-side bytes per launch of `kernel` from the COMMITTED rocprofv3 --pmc passes (profiles/<rnd>_pmc_counters.json:
    separate FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes) -- a recorded figure, not a
    measurement of this run; None when the file does not hold the kernel."""
    try:
        with open(os.path.join(ROOT, "profiles", "%s_pmc_counters.json" % rnd)) as f:
            d = json.load(f)
        v = d.get(kernel)
        return v.get("traffic_bytes") if isinstance(v, dict) else None
    except Exception:
        return None


def _profile(name):
    """Newest committed profile of that name: profiles/r06_<name> if present, else the round-5 / round-4 one."""
    for rnd in ("r06", "r05", "r04"):
        p = os.path.join(ROOT, "profiles", "%s_%s" % (rnd, name))
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", "r06_%s" % name)


def _rocprof_avg_us(kernel, full=False):
    """Average duration of `kernel` in the COMMITTED rocprofv3 --kernel-trace --stats summary of this same command
    (profiles/r04_{default,full}_bench_kernel_stats.csv).  None when the summary does not hold the kernel."""
    import csv
    try:
        path = _profile("%s_bench_kernel_stats.csv" % ("full" if full else "default"))
        with open(path) as f:
            for r in csv.DictReader(f):
                if kernel in r["Name"]:
                    return round(float(r["AverageNs"]) / 1e3, 2)
    except Exception:
        pass
    return None


def in_step_stamps(trainer, batch, dev, graph_replays=25):
    """Device time stamps captured INTO a training step's hipGraph (hip.stamp: a one-thread kernel storing the 100 MHz
    real-time counter on the stream it is launched on), both streams running: median over `graph_replays` replays, stamp
    overhead subtracted, of (a) the GIN layer's second product, (b) the last CFConv aggregation of the SchNet forward,
    (c) the tail of the step (end of the backward chain -> end of Adam and of the weight-copy refresh), (d) the step."""
    from moleculesde_amd import hip
    from moleculesde_amd.geom3d import prepare_batch
    res = {}
    try:
        b2_ = batch.clone() if hasattr(batch, "clone") else batch
        if not getattr(b2_, "_msde_plan", None):
            b2_ = prepare_batch(b2_, dev)
        hip.enable_stamps(dev)
        dp_was, trainer.dp_enabled = trainer.dp_enabled, False      # rank 0 alone: no collectives in this measurement
        trainer.step(b2_)
        trainer.capture(b2_)
        pairs = {"cf_agg": ("cf_agg_start", "cf_agg_end"), "tail": ("bwd_main_end", "step_end"), "tail_both": ("bwd_side_end", "step_end"),
                 "tail_chain": ("bwd_side_chain_end", "step_end"),
                 "step": ("step_start", "step_end")}
        for k in range(8):        # the GIN layers' second product: one pair of stamps per layer
            pairs["gin_gemm2#%d" % k] = ("gin_gemm2_start#%d" % k, "gin_gemm2_end#%d" % k)
        vals = {k: [] for k in pairs}
        for _ in range(graph_replays):
            trainer.step_graph(b2_)
            torch.cuda.synchronize()
            t = hip.read_stamps()
            for k, (s0, s1) in pairs.items():
                if s0 in t and s1 in t:
                    vals[k].append((t[s1] - t[s0]) / 100.0)          # 100 MHz counter -> us
        ovh = _stamp_overhead_us(dev)
        for k, v in vals.items():
            if v:
                v.sort()
                res[k] = v[len(v) // 2] - (ovh if (k.startswith("gin_gemm2") or k == "cf_agg") else 0.0)
        per_layer = [res[k] for k in sorted(res) if k.startswith("gin_gemm2#")]
        if per_layer:             # mean over the layers of the per-layer medians (a single layer's figure depends on what the
            res["gin_gemm2"] = sum(per_layer) / len(per_layer)      # second stream happens to run beside it)
            res["gin_gemm2_per_layer"] = [round(x, 2) for x in per_layer]
        res["replays"] = graph_replays
    except Exception as exc:
        print(f"[bench] in-step timing failed ({type(exc).__name__}: {exc})", file=sys.stderr)
    finally:
        hip.STAMPS = None
        trainer.dp_enabled = locals().get("dp_was", trainer.dp_enabled)
    return res


def in_step_family(trainer, batch, dev, graph_replays=15):
    """The WHOLE gemm_t2 family inside a captured training step: device timestamps around every launch (hip.enable_stamps(...,
    family=True): two one-thread stamp kernels per launch, both streams running), median over replays per launch, stamp
    overhead subtracted; aggregated per (N, K, A transform) class and over the family.  FLOPs = 2 M N K of the rows the batch
    really has.  The ~150 extra stamp launches lengthen the step they sit in (reported as `step_us_with_stamps`), so the
    durations are those of the kernels themselves beside a somewhat slower neighbourhood, not a statement about the step."""
    from moleculesde_amd import hip
    from moleculesde_amd.geom3d import prepare_batch
    try:
        b2_ = batch.clone() if hasattr(batch, "clone") else batch
        if not getattr(b2_, "_msde_plan", None):
            b2_ = prepare_batch(b2_, dev)
        hip.enable_stamps(dev, family=True)
        dp_was, trainer.dp_enabled = trainer.dp_enabled, False
        trainer.step(b2_)
        trainer.capture(b2_)
        shapes = list(hip.STAMPS["t2_shapes"])
        n = len(shapes)
        per = [[] for _ in range(n)]
        steps = []
        for _ in range(graph_replays):
            trainer.step_graph(b2_)
            torch.cuda.synchronize()
            t = hip.read_stamps()
            for k in range(n):
                a, b = t.get("t2_start#%d" % k), t.get("t2_end#%d" % k)
                if a is not None and b is not None:
                    per[k].append((b - a) / 100.0)
            if "step_start" in t and "step_end" in t:
                steps.append((t["step_end"] - t["step_start"]) / 100.0)
        ovh = _stamp_overhead_us(dev)
        cls = {}
        for (M, N, K, axf), v in zip(shapes, per):
            if not v:
                continue
            v.sort()
            us = max(v[len(v) // 2] - ovh, 0.1)
            c = cls.setdefault((N, K, axf), {"launches": 0, "gflop": 0.0, "us": 0.0, "M": M})
            c["launches"] += 1
            c["gflop"] += 2.0 * M * N * K / 1e9
            c["us"] += us
        tot_f = sum(c["gflop"] for c in cls.values())
        tot_us = sum(c["us"] for c in cls.values())
        rows = []
        for (N, K, axf), c in sorted(cls.items(), key=lambda kv: -kv[1]["us"]):
            tf = c["gflop"] / c["us"] * 1e3 if c["us"] else 0.0          # GFLOP per us = 1000 TFLOP/s
            rows.append({"N": N, "K": K, "a_transform": axf, "rows": c["M"], "launches_per_step": c["launches"],
                         "gflop_per_step": round(c["gflop"], 3), "us_per_step_in_step": round(c["us"], 1),
                         "achieved_tflops": round(tf, 2), "frac": round(tf / FP32_MFMA_PEAK_TF, 4)})
        steps.sort()
        return {"launches_per_step": n, "gflop_per_step": round(tot_f, 2), "us_per_step_in_step": round(tot_us, 1),
                "achieved": round(tot_f / tot_us * 1e3, 2) if tot_us else None,
                "frac": round(tot_f / tot_us * 1e3 / FP32_MFMA_PEAK_TF, 4) if tot_us else None, "classes": rows,
                "stamp_overhead_us": round(ovh, 2), "replays": graph_replays,
                "step_us_with_stamps": round(steps[len(steps) // 2], 1) if steps else None}
    except Exception as exc:
        print(f"[bench] family timing failed ({type(exc).__name__}: {exc})", file=sys.stderr)
        return None
    finally:
        hip.STAMPS = None
        trainer.dp_enabled = locals().get("dp_was", trainer.dp_enabled)


def count_kernels_per_step(trainer, batch):
    """Kernel launches of one training step (what a replayed hipGraph holds, plus the pointer-table uploads an eager step
    makes and a replay does not): counted with torch.profiler on one eager step; (launches, launches that are not kernels of
    libmsde_hip.so)."""
    try:
        from torch.profiler import profile, ProfilerActivity
        trainer.dp_enabled = False          # rank 0 measures alone: no collective may be issued here (N > 1 would deadlock)
        trainer.step(batch)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            trainer.step(batch)
            torch.cuda.synchronize()
        names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
        kern = [n for n in names if "Memcpy" not in n and "Memset" not in n]
        foreign = [n for n in kern if "at::" in n or "Cijk" in n or "elementwise" in n or "vectorized" in n]
        return len(kern), len(foreign)
    except Exception as exc:
        print(f"[bench] kernel count failed ({type(exc).__name__}: {exc})", file=sys.stderr)
        return None, None


def _rocprof_family(prefix, full=False):
    """All instantiations of a kernel template in the committed rocprofv3 summary: {share of GPU time, launches and us per
    step}; the summary covers the whole bench run, so per-step figures divide by the calls of the once-per-step Adam kernel."""
    import csv
    try:
        path = _profile("%s_bench_kernel_stats.csv" % ("full" if full else "default"))
        rows = list(csv.DictReader(open(path)))
        steps = max((int(r["Calls"]) for r in rows if "adam_chunks" in r["Name"]), default=0)
        mine = [r for r in rows if prefix in r["Name"]]
        if not mine or not steps:
            return None
        tot_ns = sum(float(r["TotalDurationNs"]) for r in mine)
        return {"share_of_gpu_time_pct": round(sum(float(r["Percentage"]) for r in mine), 2),
                "launches_per_step": round(sum(int(r["Calls"]) for r in mine) / steps, 1),
                "us_per_step_all_shapes": round(tot_ns / steps / 1e3, 1),
                "avg_launch_us_over_all_shapes": round(tot_ns / sum(int(r["Calls"]) for r in mine) / 1e3, 2)}
    except Exception:
        return None


def roofline_gemm(trainer, batch, dev, in_step, family):
    """`roofline`: the kernel FAMILY with the largest share of the step's GPU time -- gemm_t2_kernel (csrc/gemm_t2.h, the 2-D
    tiled fp32-MFMA GEMM of every node-level product: a third of the GPU time of the default step over its instantiations;
    the dominant single instantiation by the committed rocprofv3 summary is the plain <5,0,0> one).  `achieved` / `frac` = the
    FLOP-WEIGHTED figure of the whole family INSIDE the captured step: sum over its launches of 2 M N K divided by the sum of
    their durations from device timestamps around every launch, both streams running (in_step_family); `classes` lists every
    (N, K, A-transform) class with its own in-step figure.  `gin_second_product` keeps round 4's figure -- the <5,1>
    instantiation on the second product of a GIN layer ([N, 600] with BatchNorm + ReLU on the A fragments x W^T [600, 300],
    statistics in the epilogue, molecule_gnn_model.py:17,176-182), in the step and stand-alone between HIP events."""
    from moleculesde_amd import hip
    N = int(batch.x.size(0))
    D, H = trainer.args.emb_dim, 2 * trainer.args.emb_dim
    flops = 2.0 * N * D * H
    sub = {"kernel": "gemm_t2_kernel<5, 1> (GIN layer, second product: BatchNorm+ReLU on the A fragments, statistics in the "
                     "epilogue)", "flops_per_launch": flops, "shape_MxNxK": [N, D, H]}
    # ---- stand-alone: the same fused launch between HIP events
    with torch.no_grad():
        z1 = torch.randn(N, H, device=dev)
        W = torch.randn(D, H, device=dev) / H ** 0.5
        sc, sh = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev) * 0.1
        b2 = torch.randn(D, device=dev)
        a1, z2 = torch.empty(N, H, device=dev), torch.empty(N, D, device=dev)
        strips, _ = hip.rs_geometry(N, D, H)
        st = torch.empty(strips, 2, D, device=dev)
        fn = lambda: hip.gemm_node(z1, W, z2, True, D, H, bias=b2, axf="affine", xf=(sc, sh), relu=True, A_out=a1, stats=st,
                                   stats_mode="bnfwd")
        ms = _event_time_ms(fn, 50, torch.cuda.current_stream())
    tf = flops / (ms * 1e-3) / 1e12
    sub["standalone"] = {"avg_launch_us": round(ms * 1e3, 2), "achieved": round(tf, 2), "frac": round(tf / FP32_MFMA_PEAK_TF, 4)}
    in_us = in_step.get("gin_gemm2")
    if in_us and in_us > 0:
        tfi = flops / (in_us * 1e-6) / 1e12
        sub.update({"achieved": round(tfi, 2), "frac": round(tfi / FP32_MFMA_PEAK_TF, 4), "avg_launch_us": round(in_us, 2),
                    "timing": "inside the captured step (device timestamps around the launch in each of the 5 GIN layers, "
                              "median of %d replays per layer, mean over the layers, stamp overhead subtracted)"
                              % in_step.get("replays", 0),
                    "avg_launch_us_per_layer": in_step.get("gin_gemm2_per_layer")})
    sub["traffic"] = _pmc_traffic("gemm_t2_kernel<5, 1>[3588x300x600]", "r04")
    sub["traffic_source"] = "profiles/r04_pmc_counters.json (committed FETCH_SIZE / WRITE_SIZE passes; not measured by this run)"
    sub["algorithmic_bytes_per_launch"] = (N * H + H * D + N * D + N * H) * 4      # z1 in, W, z2 out, a1 out
    out = {"kernel": "gemm_t2_kernel<RN, AXF> family (every node-level product of the step), FLOP-weighted, in the step",
           "bound": "mfma", "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s"}
    if family and family.get("frac"):
        out.update({"achieved": family["achieved"], "frac": family["frac"],
                    "timing": "device timestamps around every launch of the family inside the captured two-stream step, median of "
                              "%d replays per launch, stamp overhead (%.2f us) subtracted" % (family["replays"], family["stamp_overhead_us"]),
                    "launches_per_step": family["launches_per_step"], "gflop_per_step": family["gflop_per_step"],
                    "us_per_step_in_step": family["us_per_step_in_step"], "classes": family["classes"],
                    "step_us_with_stamps": family["step_us_with_stamps"]})
    else:
        out.update({"achieved": sub.get("achieved", sub["standalone"]["achieved"]), "frac": sub.get("frac", sub["standalone"]["frac"]),
                    "timing": "family timing unavailable: the GIN second product alone"})
    fam = _rocprof_family("gemm_t2_kernel")
    if fam:
        out["family_in_committed_rocprofv3_summary"] = fam
    # counters of the dominant instantiations (separate --pmc passes, tools/gpu_r05_pmc.sh -> profiles/r05_pmc_counters.json)
    out["traffic"] = _pmc_traffic("gemm_t2_kernel<5, 0>[3588x300x300]", "r06") or _pmc_traffic("gemm_t2_kernel<5, 0>[3588x300x300]", "r05") or sub["traffic"]
    out["traffic_source"] = "profiles/r06_pmc_counters.json (committed FETCH_SIZE / WRITE_SIZE passes of the plain N = K = 300 product inside ten eager steps; not measured by this run)"
    out["gin_second_product"] = sub
    return out


def _stamp_overhead_us(dev):
    """Two back-to-back timestamp launches on one stream: the cost the stamp pair itself adds between its two reads."""
    from moleculesde_amd import hip
    g = torch.cuda.CUDAGraph()
    hip.stamp("ovh_a"); hip.stamp("ovh_b")
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        hip.stamp("ovh_a")
        hip.stamp("ovh_b")
    v = []
    for _ in range(20):
        g.replay()
        torch.cuda.synchronize()
        t = hip.read_stamps()
        v.append((t["ovh_b"] - t["ovh_a"]) / 100.0)
    v.sort()
    return v[len(v) // 2]


def _pair_setup(trainer, batch):
    from moleculesde_amd import hip, plan as P
    sch = trainer.models["model_3D"]
    pl = P.get_plan(batch)
    with torch.no_grad():
        pp = hip.pair_plan(batch.positions, pl, sch.cutoff)
    return sch, pl, pp, int(pp.count[0]), int(batch.x.size(0))


def roofline_pair_filter(trainer, batch, iters=50):
    """CFConv filter network on unordered pairs (csrc/cfconv_pair.hip; 6 launches per forward): FLOPs per launch =
    P * 2 * (G F + F F) over the P = E_r / 2 pairs -- HALF of the per-edge kernel of rounds 1-2 (2.25 GFLOP at E_r = 49 k),
    which is the point: the fraction is reported against the work actually done AND against the per-edge figure."""
    from moleculesde_amd import hip, _lib
    sch, pl, pp, P2, N = _pair_setup(trainer, batch)
    if sch.num_filters != 128:
        return None
    blk, de = sch.interactions[0], sch.distance_expansion
    G = sch.num_gaussians
    with torch.no_grad():
        W1, b1, W2, b2 = (t.detach() for t in (blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight, blk.mlp[2].bias))
        Wf = torch.empty(max(pp.P, 1), 128, device=batch.x.device)
        p, st = hip._p, hip._stream()
        fn = lambda: _lib.call("msde_cfconv_pair_filter", p(pp.pd), p(pp.count), p(W1), p(b1), p(W2), p(b2), p(de.offset), 128, G,
                               pp.P, float(de.coeff), float(sch.cutoff), 0, p(Wf), st)
        ms = _event_time_ms(fn, iters, torch.cuda.current_stream())
    flops = P2 * 2.0 * (G * 128 + 128 * 128)
    tf = flops / (ms * 1e-3) / 1e12
    multi = None
    try:      # round 6: the rows of ALL interaction blocks from one launch (what the step runs: hip.cfconv_pair_filters)
        nets = [(b_.mlp[0].weight.detach(), b_.mlp[0].bias.detach(), b_.mlp[2].weight.detach(), b_.mlp[2].bias.detach())
                for b_ in sch.interactions]
        with torch.no_grad():
            ms_m = _event_time_ms(lambda: hip.cfconv_pair_filters(pp, nets, de.offset, de.coeff, sch.cutoff), iters,
                                  torch.cuda.current_stream())
        tf_m = flops * len(nets) / (ms_m * 1e-3) / 1e12
        multi = {"kernel": "cfconv_pair_filter_multi_kernel", "blocks": len(nets), "avg_launch_us": round(ms_m * 1e3, 2),
                 "us_per_interaction_block": round(ms_m * 1e3 / len(nets), 2), "achieved": round(tf_m, 2),
                 "frac": round(tf_m / FP32_MFMA_PEAK_TF, 4), "timing": "stand-alone (HIP events), incl. the [L, P, 128] allocation"}
    except Exception as exc:
        multi = {"error": f"{type(exc).__name__}: {exc}"}
    return {"kernel": "cfconv_pair_filter_kernel", "bound": "mfma", "achieved": round(tf, 2), "peak": FP32_MFMA_PEAK_TF,
            "all_blocks_one_launch": multi,
            "unit": "TFLOP/s", "frac": round(tf / FP32_MFMA_PEAK_TF, 4), "flops_per_launch": flops, "pairs": P2,
            "frac_against_the_per_edge_flops_of_rounds_1_2": round(2 * tf / FP32_MFMA_PEAK_TF, 4),
            "avg_launch_us": round(ms * 1e3, 2), "launches_per_step": 6, "timing": "stand-alone (HIP events)",
            "rocprofv3_avg_launch_us_committed_profile": _rocprof_avg_us("cfconv_pair_filter_kernel"),
            "algorithmic_bytes_per_launch": P2 * 4 + P2 * 128 * 4 + (G * 128 + 128 * 128 + 256) * 4}


def roofline_pair_bwd_w(trainer, batch, iters=30):
    """Pair form of the recomputing weight-gradient kernel (cfconv_fused_bwd.hip, SYM): FLOPs per launch =
    P * 2 * (2 G F + 2 F F) over unordered pairs (4.5 GFLOP per launch in the per-edge form of rounds 1-2)."""
    from moleculesde_amd import hip, _lib
    sch, pl, pp, P2, N = _pair_setup(trainer, batch)
    if sch.num_filters != 128:
        return None
    blk, de = sch.interactions[0], sch.distance_expansion
    G = sch.num_gaussians
    with torch.no_grad():
        dev = batch.x.device
        x1, g = torch.randn(N, 128, device=dev), torch.randn(N, 128, device=dev)
        W1, b1, W2 = blk.mlp[0].weight.detach(), blk.mlp[0].bias.detach(), blk.mlp[2].weight.detach()
        ws = hip._cf_workspace(pp.P, G, dev, 0)
        p, st = hip._p, hip._stream()
        fn = lambda: _lib.call("msde_cfconv_pair_bwd_w", p(g), p(x1), p(pp.pd), p(pp.count), p(pp.pi), p(pp.pj), p(W1), p(b1), p(W2),
                               p(de.offset), N, 128, G, pp.P, float(de.coeff), float(sch.cutoff), 0, p(None), p(None), p(None),
                               p(None), p(ws), st)
        ms = _event_time_ms(fn, iters, torch.cuda.current_stream())
    flops = P2 * 2.0 * (2 * G * 128 + 2 * 128 * 128)
    tf = flops / (ms * 1e-3) / 1e12
    return {"kernel": "cfconv_fused_bwd_w_pipe_kernel<26, 0, true, true> (pair form)", "bound": "mfma", "achieved": round(tf, 2),
            "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / FP32_MFMA_PEAK_TF, 4), "flops_per_launch": flops,
            "frac_against_the_per_edge_flops_of_rounds_1_2": round(2 * tf / FP32_MFMA_PEAK_TF, 4), "pairs": P2,
            "avg_launch_us": round(ms * 1e3, 2), "launches_per_step": 6, "timing": "stand-alone (HIP events)",
            "rocprofv3_avg_launch_us_committed_profile": _rocprof_avg_us("cfconv_fused_bwd_w_pipe_kernel<26, 0, true, true>")}


def roofline_hbm_kernel(trainer, batch, in_step, iters=50):
    """HBM-bound message-passing kernel of the step (12 launches: CFConv aggregation forward and its input gradient):
    out[i] = sum_{j != i} x[j] * Wf[pair(i, j)] (schnet.py:190,194-195; csrc/cfconv_pair.hip).  Algorithmic bytes per launch
    (SURVEY §8d convention, a gather counts E x row bytes): E F 4 (gathered x rows) + E F 4 (filter rows, each pair row read
    by both of its atoms) + N F 4 (output) + index arrays, with E = 2 P ordered neighbours.
    `achieved` / `frac` use the launch's duration INSIDE the captured step (device stamps; the other stream's kernels share
    the chip).  Back-to-back launches on one resident 52 MB working set are served by L2 / Infinity Cache and read above the
    HBM peak: that figure is kept under `standalone_cache_resident` and is not a roofline fraction.  `counter_bytes`: what
    the FETCH_SIZE / WRITE_SIZE passes of the committed profile saw leave the L2 for the same launch."""
    from moleculesde_amd import hip, _lib
    sch, pl, pp, P2, N = _pair_setup(trainer, batch)
    Fd = sch.num_filters
    if Fd != 128:
        return None
    with torch.no_grad():
        dev = batch.x.device
        x = torch.randn(N, Fd, device=dev)
        Wf = torch.randn(max(pp.P, 1), Fd, device=dev)
        out = torch.empty(N, Fd, device=dev)
        p, st = hip._p, hip._stream()
        fn = lambda: _lib.call("msde_cfconv_pair_aggregate", p(x), p(Wf), p(pp.batch_i32), p(pp.mol_ptr), p(pp.pair_ptr), N, pp.B,
                               Fd, p(out), st)
        ms = _event_time_ms(fn, iters, torch.cuda.current_stream())
    E = 2 * P2
    nbytes = E * Fd * 4 * 2 + N * Fd * 4 + N * 4 + (pp.B + 1) * 8
    alone = nbytes / (ms * 1e-3) / 1e9
    traffic = _pmc_traffic("cfconv_pair_aggregate_kernel")
    res = {"kernel": "cfconv_pair_aggregate_kernel", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "bytes_per_launch": nbytes, "traffic": traffic, "launches_per_step": 12,
           "standalone_cache_resident": {"avg_launch_us": round(ms * 1e3, 2), "algorithmic_GBps": round(alone, 1),
                                         "note": "back-to-back launches on one resident working set: served by L2 / Infinity "
                                                 "Cache, NOT an HBM figure"},
           "rocprofv3_avg_launch_us_committed_profile": _rocprof_avg_us("cfconv_pair_aggregate_kernel")}
    us = in_step.get("cf_agg")
    timing = "inside the captured step (device timestamps around the launch, median of %d replays)" % in_step.get("replays", 0)
    if not us or us <= 0:          # fall back to the committed rocprofv3 in-step average
        us = res["rocprofv3_avg_launch_us_committed_profile"]
        timing = "committed rocprofv3 average of this kernel inside the bench (live stamps unavailable)"
    if us:
        g = nbytes / (us * 1e-6) / 1e9
        res.update({"avg_launch_us": round(us, 2), "achieved": round(g, 1), "frac": round(min(g / HBM_PEAK_GBS, 1.0), 4),
                    "timing": timing})
        if traffic:
            gc = traffic / (us * 1e-6) / 1e9
            res["counter_bytes"] = {"achieved": round(gc, 1), "frac": round(gc / HBM_PEAK_GBS, 4)}
    return res


def forward_algorithmic(st, H=300, F=128, G=51, D=300, C=32, L3=6):
    """Algorithmic bytes and FLOPs of the SchNet + SDEModel2Dto3D_02 FORWARD for a batch with the counts `st`
    (formulas not constants: fp32 = 4 B, index = 4 B, each logical op reads each distinct input
    once and writes its output once, a gather counts E x row bytes, temporaries inside one logical op are free)."""
    N, E_e, E_r = st["N"], st["E_e"], st["E_r"]
    b, f = {}, {}
    b["schnet_embedding"] = N * 4 + N * H * 4
    b["schnet_radius"] = N * 16 + E_r * 12
    b["schnet_lin1"] = L3 * (N * H + H * F + N * F) * 4
    b["schnet_cfconv"] = L3 * (E_r * 8 + E_r * F * 4 + (G * F + F + F * F + F) * 4 + N * F * 4 + (N + 1) * 4)
    b["schnet_lin2_ssp_lin_res"] = L3 * (N * F * 4 + 2 * N * H * 4 + (F * H + H + H * H + H) * 4)
    b["schnet_head"] = 2 * N * H * 4 + 2 * (H * H + H) * 4
    b["sde_edge_2D_emb"] = 3 * N * D * 4 + 2 * D * D * 4 + E_e * 8 + 2 * E_e * D * 4 + E_e * C * 4 + D * C * 4
    b["sde_edge_geometry"] = E_e * (8 + 24 + 2 * C * 4 + 36)
    b["sde_node_emb"] = N * D * 4 + N * C * 4 + D * C * 4
    b["sde_gat_layers"] = 4 * (8 * N * C * 4 + E_e * 8 + 3 * E_e * C * 4)
    b["sde_basis_mlp_mean"] = 2 * (E_e * 8 + 3 * E_e * C * 4 + E_e * 36 + N * 12)
    f["schnet_lin1"] = L3 * 2 * N * H * F
    f["schnet_cfconv"] = L3 * E_r * (2 * (G * F + F * F) + 5 * F + 5 * G)
    f["schnet_lin2_ssp_lin_res"] = L3 * (2 * N * F * H + 2 * N * H * H + 4 * N * H)
    f["schnet_head"] = 2 * 2 * N * H * H
    f["sde_edge_2D_emb"] = 2 * N * D * 2 * D + 4 * E_e * D + 2 * E_e * D * C
    f["sde_edge_geometry"] = E_e * (2 * 2 * C * C + 2 * 2 * 4 * C * C + 2 * (2 * C + 2) * C + 2 * C * C + 400)
    f["sde_node_emb"] = 2 * N * D * C
    f["sde_gat_layers"] = 4 * (N * (2 * C * 4 * C + 2 * 2 * C * C) + E_e * (2 * C * C + 8 * C))
    f["sde_basis_mlp_mean"] = 2 * E_e * (2 * 2 * C * 128 + 2 * 128 * 3 + 50)
    return b, f


def roofline_forward(trainer, batch, stats, iters=30):
    """The north-star figure (BASELINE.json; SURVEY §8d 'SchNet + 2D->3D fwd'): the training-mode FORWARD of SchNet
    and SDEModel2Dto3D_02 at this batch, captured as ONE hipGraph (both models back to back on one stream, tape
    built as in a training step, 2D representation = a resident [N, emb] tensor) and replayed `iters` times between
    HIP events.  Reported against both bounds: algorithmic bytes / 8 TB/s and algorithmic FLOPs / 157.3 TFLOP/s."""
    m = trainer.models
    sch, sde = m["model_3D"], m["SDE_2Dto3D_model"]
    dev = batch.x.device
    h2 = torch.randn(batch.x.size(0), trainer.args.emb_dim, device=dev, requires_grad=True)
    keep = []

    side = trainer._side_stream

    def fwd():
        # the two models are independent until the losses: SchNet runs on the trainer's second stream beside the 2D->3D
        # model, exactly as in a training step (pretrain.Trainer.losses)
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            _, h3 = sch(batch.x[:, 0], batch.positions, batch.batch, return_latent=True)
        l23 = sde(h2, batch, anneal_power=0)["position"]
        main.wait_stream(side)
        h3.record_stream(main)
        keep[:] = [h3, l23]

    for _ in range(2):
        fwd()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        from moleculesde_amd.slabs import no_gc
        with no_gc(), torch.cuda.graph(g, capture_error_mode="thread_local"):
            fwd()
        run = g.replay
        how = "hipGraph replay"
    except Exception as exc:
        print(f"[bench] forward capture failed ({type(exc).__name__}: {exc}); timing eager", file=sys.stderr)
        run, how = fwd, "eager"
    torch.cuda.synchronize()
    ms = _event_time_ms(run, iters, torch.cuda.current_stream())
    by, fl = forward_algorithmic(stats, H=trainer.args.emb_dim, D=trainer.args.emb_dim)
    nbytes, flops = float(sum(by.values())), float(sum(fl.values()))
    # what the kernels EXECUTE: the filter network of CFConv runs once per unordered pair (csrc/cfconv_pair.hip), i.e. on
    # E_r / 2 rows instead of the E_r rows the algorithmic count (the reference's per-edge form) charges
    fl_exec = dict(fl)
    F_, G_, L3_ = 128, 51, 6
    fl_exec["schnet_cfconv"] = L3_ * ((stats["E_r"] / 2.0) * (2 * (G_ * F_ + F_ * F_) + 5 * F_ + 5 * G_) + stats["E_r"] * 2 * F_)
    flops_exec = float(sum(fl_exec.values()))
    B = stats["B"]
    t = ms * 1e-3
    gbs, tf = nbytes / t / 1e9, flops / t / 1e12
    return {"what": "SchNet + SDEModel2Dto3D_02 forward, training mode, bs %d (%s; SchNet on the second stream beside the "
                    "2D->3D model, as in a training step)" % (B, how), "ms": round(ms, 4),
            "molecules_per_s_forward_only": round(B / t, 1),
            "algorithmic_MB": round(nbytes / 1e6, 2), "algorithmic_MB_per_molecule": round(nbytes / 1e6 / B, 4),
            "algorithmic_GFLOP": round(flops / 1e9, 3), "algorithmic_MFLOP_per_molecule": round(flops / 1e6 / B, 2),
            "hbm": {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)},
            "fp32_flop_floor": {"achieved": round(tf, 2), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                "frac": round(tf / FP32_MFMA_PEAK_TF, 4)},
            "executed_GFLOP": round(flops_exec / 1e9, 3),
            "executed": {"achieved": round(flops_exec / t / 1e12, 2), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                         "frac": round(flops_exec / t / 1e12 / FP32_MFMA_PEAK_TF, 4),
                         "note": "CFConv filter network on unordered pairs: half the per-edge FLOPs of the algorithmic count"},
            "us_at_100pct_hbm": round(nbytes / (HBM_PEAK_GBS * 1e9) * 1e6, 1),
            "us_at_fp32_peak": round(flops / (FP32_MFMA_PEAK_TF * 1e12) * 1e6, 1)}


def roofline_dense_head_node_mlp(batch, iters=20):
    """MFMA utilisation of the dense head's node MLP (NodeScoreNetwork_dense.final: 364 -> 728 -> 728 -> 119 with SiLU,
    invariant_scorenetwork_dense.py:126-127; 2/3 of the head's FLOPs) as the product runs it: three msde_gemm_ex launches
    (csrc/gemm_ex.hip, bias + SiLU + pre-activation store fused) over the VALID atoms only (no padding to B*N_max).
    FLOPs = 2 * N * (364*728 + 728*728 + 728*119); timed as a captured hipGraph of the three launches."""
    from moleculesde_amd import hip
    dev = batch.x.device
    N = int(batch.x.size(0))
    with torch.no_grad():
        X = torch.randn(N, 364, device=dev)
        W = [torch.randn(728, 364, device=dev) / 19, torch.randn(728, 728, device=dev) / 27, torch.randn(119, 728, device=dev) / 27]
        b = [torch.randn(728, device=dev), torch.randn(728, device=dev), torch.randn(119, device=dev)]
        Z1, F1, Z2, F2 = (torch.empty(N, 728, device=dev) for _ in range(4))
        OUT = torch.empty(N, 120, device=dev)

        from moleculesde_amd.geom3d import dense_head as DH
        W = [torch.nn.Parameter(w) for w in W]
        b = [torch.nn.Parameter(v) for v in b]

        def chain():            # exactly the launches of dense_head.node_forward
            DH.node_mlp_forward(X, W[0], b[0], W[1], b[1], W[2], b[2], Z1, F1, Z2, F2, OUT)
        for _ in range(3):
            chain()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            chain()
        ms = _event_time_ms(g.replay, iters, torch.cuda.current_stream())
    flops = 2.0 * N * (364 * 728 + 728 * 728 + 728 * 119)
    tf = flops / (ms * 1e-3) / 1e12
    return {"kernel": "gemm_ex_kernel (364->728) + gemm_t2_kernel (728->728) + gemm_rsa_kernel (728->119): node MLP of the dense head, valid atoms only", "bound": "mfma",
            "achieved": round(tf, 2), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / FP32_MFMA_PEAK_TF, 4),
            "us_per_chain": round(ms * 1e3, 2), "rows": N, "flops": flops}


def config4_sampler(dev, pc_steps=1000, cpu_steps=10):
    """BASELINE.json configs[3]: 2D->3D VE reverse-SDE sampling (pretrain_MoleculeSDE_inference_2D_to_3D_VE_VP.py:92-138 as
    intended): one 14-atom molecule x 10 replicas, `pc_steps` predictor-corrector iterations (2 score-network
    calls each), one iteration captured as a hipGraph and replayed.  Beside it the oracle's models through the same loop on
    the host cores, on a bounded sample of `cpu_steps` iterations."""
    import numpy as np
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import sampler
    from moleculesde_amd.batch import Batch
    from moleculesde_amd.synthetic import make_molecule
    torch.manual_seed(0)
    mol = make_molecule(np.random.default_rng(0), 14)
    cpu_b = Batch.from_data_list([mol] * 10)
    b = G.prepare_batch(cpu_b.clone(), dev)
    gnn = G.GNN(5, 300, JK="last", drop_ratio=0, gnn_type="GIN").to(dev).eval()
    s23 = G.SDEModel2Dto3D_02(emb_dim=300, hidden_dim=32, beta_min=0.2, beta_max=1.0, num_diffusion_timesteps=1000,
                              beta_schedule=None, SDE_type="VE", use_extend_graph=True).to(dev).eval()
    with torch.no_grad():
        rep = gnn(b.x, b.edge_index, b.edge_attr)
    sampler.position_PC_generation(s23, rep, b, num_steps=20, use_graph=True)      # warm-up (workspaces, first capture)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pos = sampler.position_PC_generation(s23, rep, b, num_steps=pc_steps, use_graph=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"what": "2D->3D VE predictor-corrector sampling, 10 x 14 atoms, %d PC iterations (2 score calls each)" % pc_steps,
           "seconds_per_trajectory_batch": round(dt, 3), "pc_iterations_per_s": round(pc_steps / dt, 1),
           "finite": bool(torch.isfinite(pos).all()), "launch": "one PC iteration captured as a hipGraph, replayed"}
    try:
        from oracle import restate as R
        torch.set_num_threads(min(_host_cores()[0], 16))
        ognn = R.GNN(5, 300).eval()
        o23 = R.SDEModel2Dto3D_02(emb_dim=300, hidden_dim=32, beta_schedule=None, beta_min=0.2, beta_max=1.0,
                                  num_diffusion_timesteps=1000, SDE_type="VE", use_extend_graph=True).eval()
        with torch.no_grad():
            orep = ognn(cpu_b.x, cpu_b.edge_index, cpu_b.edge_attr)
            sampler.position_PC_generation(o23, orep, cpu_b, num_steps=2, use_graph=False)
            t0 = time.perf_counter()
            sampler.position_PC_generation(o23, orep, cpu_b, num_steps=cpu_steps, use_graph=False)
            dtc = time.perf_counter() - t0
        out["cpu_oracle"] = {"pc_iterations_per_s": round(cpu_steps / dtc, 2), "sample": "%d PC iterations of the same loop on "
                             "oracle/restate.py models, %d threads" % (cpu_steps, torch.get_num_threads()),
                             "seconds_per_trajectory_batch_extrapolated": round(dtc / cpu_steps * pc_steps, 1)}
    except Exception as exc:
        out["cpu_oracle"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def config5_md17(dev, steps=20, cpu_steps=5):
    """BASELINE.json configs[4]: MD17-aspirin-shaped force fine-tuning step (finetune_MD17.py:47-78): 21 atoms, batch 1,
    SchNet(300, 128 filters, 6 interactions, 51 Gaussians, cutoff 10) + Linear head; energy -> forces by
    autograd.grad(create_graph=True) -> L1 losses -> backward through the forces -> Adam, through
    moleculesde_amd.finetune_md17.ForceTrainer: the whole step as ONE replayed hipGraph (`ms_per_step`) and launched from
    the host (`eager_ms_per_step`: ~1500 tiny launches, latency bound).  Beside it the oracle on the host cores."""
    import moleculesde_amd.geom3d as G
    from moleculesde_amd.synthetic import make_md17_batch
    kw = dict(hidden_channels=300, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=10, readout="mean", node_class=119)
    torch.manual_seed(0)
    cpu_b = make_md17_batch(1, seed=3, n_atoms=21)
    e_t, f_t = torch.randn(1, 1), torch.randn(cpu_b.x.size(0), 3)

    def step(model, head, opt, b, et, ft):
        pos = b.positions.clone().requires_grad_(True)
        energy = head(model(b.x, pos, b.batch))
        force = -torch.autograd.grad(energy, pos, grad_outputs=torch.ones_like(energy), create_graph=True, retain_graph=True)[0]
        loss = (energy - et).abs().mean() + (force - ft).abs().mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    from moleculesde_amd.finetune_md17 import ForceTrainer
    sch, head = G.SchNet(**kw).to(dev), torch.nn.Linear(300, 1).to(dev)
    b = G.prepare_batch(cpu_b.clone(), dev)
    ft = ForceTrainer(sch, head, lr=5e-4, energy_coeff=1.0, force_coeff=1.0)
    et, ftg = e_t.to(dev).view(-1), f_t.to(dev)
    for _ in range(3):
        ft.step(b, et, ftg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = ft.step(b, et, ftg)
    torch.cuda.synchronize()
    dt_eager = (time.perf_counter() - t0) / steps
    out = {"what": "MD17-aspirin-shaped SchNet force fine-tune step, 21 atoms, batch 1 (energy, forces with create_graph, "
                   "backward through the forces, Adam): moleculesde_amd.finetune_md17.ForceTrainer",
           "steps": steps, "eager_ms_per_step": round(dt_eager * 1e3, 2)}
    try:
        ft.capture(b, et, ftg)
        g = torch.Generator().manual_seed(1)
        confs = [(cpu_b.positions + 0.05 * torch.randn(cpu_b.positions.shape, generator=g)).to(dev) for _ in range(8)]
        for i in range(4):
            ft.step_graph(confs[i % 8], et, ftg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10 * steps):
            loss = ft.step_graph(confs[i % 8], et, ftg)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (10 * steps)
        out.update({"ms_per_step": round(dt * 1e3, 3), "launch": "ONE hipGraph for the whole step, replayed on 8 conformations "
                    "(positions / energies / forces copied into static buffers)", "timed_steps": 10 * steps})
    except Exception as exc:
        print(f"[bench] MD17 capture failed ({type(exc).__name__}: {exc}); reporting eager", file=sys.stderr)
        out.update({"ms_per_step": round(dt_eager * 1e3, 2), "launch": "eager"})
    out["finite"] = bool(torch.isfinite(loss))
    try:
        from oracle import restate as R
        torch.set_num_threads(min(_host_cores()[0], 16))
        osch, ohead = R.SchNet(**kw), torch.nn.Linear(300, 1)
        oopt = torch.optim.Adam(list(osch.parameters()) + list(ohead.parameters()), lr=5e-4)
        step(osch, ohead, oopt, cpu_b, e_t, f_t)
        t0 = time.perf_counter()
        for _ in range(cpu_steps):
            step(osch, ohead, oopt, cpu_b, e_t, f_t)
        out["cpu_oracle"] = {"ms_per_step": round((time.perf_counter() - t0) / cpu_steps * 1e3, 1),
                             "sample": "%d steps of oracle/restate.py SchNet, %d threads" % (cpu_steps, torch.get_num_threads())}
    except Exception as exc:
        out["cpu_oracle"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def _host_cores():
    logical = os.cpu_count() or 1
    try:          # physical cores = distinct (package, core) pairs
        phys = set()
        pk = co = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pk = line.split(":")[1].strip()
            elif line.startswith("core id"):
                co = line.split(":")[1].strip()
            elif not line.strip():
                if pk is not None and co is not None:
                    phys.add((pk, co))
                pk = co = None
        cores = len(phys) or logical
        model = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")
    except Exception:
        cores, model = logical, "?"
    return cores, logical, model


def _cpu_steps(threads, bs, warm, timed, budget_s):
    from oracle import restate as R
    from moleculesde_amd.synthetic import make_batch
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    models = R.build_models(use_3d2d=False)
    opt = R.make_optimizer(models, lr=1e-4, gnn_2d_lr_scale=1.0, gnn_3d_lr_scale=0.1)
    b = make_batch(bs, seed=0)
    times, stages = [], []
    spent = 0.0
    for i in range(warm + timed):
        t = [time.perf_counter()]
        h2 = models["model_2D"](b.x, b.edge_index, b.edge_attr); t.append(time.perf_counter())
        _, h3 = models["model_3D"](b.x[:, 0], b.positions, b.batch, return_latent=True); t.append(time.perf_counter())
        cl, _ = R.dual_CL(h2, h3, 0.1); t.append(time.perf_counter())
        l23 = models["SDE_2Dto3D_model"](h2, b, anneal_power=0)["position"]; t.append(time.perf_counter())
        loss = cl + l23
        opt.zero_grad()
        loss.backward(); t.append(time.perf_counter())
        opt.step(); t.append(time.perf_counter())
        dt = t[-1] - t[0]
        print(f"[cpu_baseline] {threads} threads, step {i}: {dt:.2f}s", file=sys.stderr, flush=True)
        if i >= warm:
            times.append(dt)
            stages.append([t[k + 1] - t[k] for k in range(6)])
            spent += dt
            if spent > budget_s:
                break
    order = sorted(range(len(times)), key=lambda k: times[k])
    mid = order[len(order) // 2]
    names = ["GIN_fwd", "SchNet_fwd", "contrastive_fwd", "SDE2Dto3D_fwd", "backward", "Adam"]
    return times[mid], len(times), {n: round(v * 1e3, 1) for n, v in zip(names, stages[mid])}


def cpu_baseline(bs=256):
    """The oracle port of the same step (oracle/restate.py, plain PyTorch fp32 on the host cores) on a
    bounded sample.  `value` = the better of two thread counts, both reported: 16 threads (3 warm-up + 10 timed steps,
    median; thousands of small eager operators per step -- beyond ~16 threads the per-operator barrier dominates) and
    ALL physical cores (1 + 3 steps).  `cores` = the thread count `value` was measured with.  Per-stage split of the
    median step included."""
    cores, logical, model = _host_cores()
    t16 = min(cores, 16)
    med16, n16, st16 = _cpu_steps(t16, bs, 3, 10, 20.0)
    out = {"value": round(bs / med16, 1), "unit": "molecules/s", "cores": t16, "kind": "port",
           "sample": f"{n16} timed steps (median) of one bs-{bs} synthetic batch after 3 warm-up, oracle/restate.py on "
                     f"torch CPU fp32, {t16} threads of {cores} physical cores ({logical} logical) of {model}",
           "ms_per_step": round(med16 * 1e3, 1), "stage_ms_median_step": st16}
    if cores > t16:
        meda, na, sta = _cpu_steps(cores, bs, 1, 3, 12.0)
        out["all_physical_cores"] = {"threads": cores, "value": round(bs / meda, 1), "ms_per_step": round(meda * 1e3, 1),
                                     "timed_steps": na, "stage_ms_median_step": sta}
        if meda < med16:
            out.update({"value": round(bs / meda, 1), "cores": cores, "ms_per_step": round(meda * 1e3, 1)})
    return out


# Keys every `--gpus N` line carries beside the contract's own (tests/test_host_logic.py asserts them on the launcher path,
# tools/dp2_gloo_smoke.sh on a real step): configs[2]'s full step timed by the same ranks, the pieces of a DP step, the
# per-rank spread.
DP_LINE_KEYS = ("config2_full", "dp_step_parts_us", "ms_per_step_per_rank", "rccl_ranks_seen", "rank_devices", "dp_backend")
DP_PART_KEYS = ("graph_us", "allreduce_us", "adam_us", "allreduce_alone_us", "steps")


def dp_step_parts(trainer, step_i, n=20):
    """The pieces of a data-parallel step, from HIP events on the compute stream (median over `n` steps, this rank):
    `graph_us` = the captured graph (plan, forward, backward, weight gradients, gradient flattening); `allreduce_us` = what the
    compute stream WAITS for the bucketed collectives (they run on RCCL's stream; bucket k+1 is on the wire while bucket k's
    Adam runs, so this is the exposed part); `adam_us` = the per-bucket optimiser launches; `allreduce_alone_us` = the same
    bucketed collectives issued on an otherwise idle device.  Every rank runs this (collectives inside)."""
    import statistics
    from moleculesde_amd import dp
    try:
        for i in range(3):
            step_i(i)
        trainer.dp_timing = []
        for i in range(n):
            step_i(i)
        torch.cuda.synchronize()
        recs, trainer.dp_timing = trainer.dp_timing, None
        g_us, ar_us, ad_us = [], [], []
        for ev in recs:
            if len(ev) < 4:
                continue
            g_us.append(ev[0].elapsed_time(ev[1]) * 1e3)
            ar = ad = 0.0
            for k in range(2, len(ev), 2):
                ar += ev[k - 1].elapsed_time(ev[k]) * 1e3
                ad += ev[k].elapsed_time(ev[k + 1]) * 1e3
            ar_us.append(ar)
            ad_us.append(ad)
        alone = []
        flat, ranges = trainer.opt.flat_g, trainer.opt.bucket_ranges
        for _ in range(max(n // 2, 3)):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _, works, _ = dp.allreduce_buckets_async(flat, ranges)
            for w in works:
                if w is not None:
                    w.wait()
            e1.record()
            e1.synchronize()
            alone.append(e0.elapsed_time(e1) * 1e3)
        flat.zero_()
        med = lambda v: round(statistics.median(v), 1) if v else None
        return {"graph_us": med(g_us), "allreduce_us": med(ar_us), "adam_us": med(ad_us), "allreduce_alone_us": med(alone),
                "steps": len(g_us), "gradient_bytes": int(flat.numel() * 4),
                "buckets_bytes": [int((b - a_) * 4) for a_, b in ranges],
                "what": "HIP events on the compute stream, median per step on this rank: graph_us = the captured step graph, "
                        "allreduce_us = what the compute stream waits for the per-model gradient buckets (exposed part; the next "
                        "bucket is on the wire while a bucket's Adam runs), adam_us = the per-bucket Adam launches, "
                        "allreduce_alone_us = the same collectives on an otherwise idle device"}
    except Exception as exc:
        trainer.dp_timing = None
        print(f"[bench] DP step parts failed ({type(exc).__name__}: {exc})", file=sys.stderr)
        return {"error": f"{type(exc).__name__}: {exc}"}


def gather_ranks(vec, device):
    """Every rank's row of floats (None -> nan), as a list of lists on every rank."""
    t = torch.tensor([float("nan") if v is None else float(v) for v in vec], device=device, dtype=torch.float64)
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return [t.tolist()]
    got = [torch.zeros_like(t) for _ in range(torch.distributed.get_world_size())]
    torch.distributed.all_gather(got, t)
    return [g.tolist() for g in got]


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n, argv, timeout=None):
    """`bench.py --gpus N` without torchrun's environment: start `python -m torch.distributed.run --nproc-per-node N
    bench.py <same flags>` as a child process (never exec: SURVEY §8e / the pool's rule about processes that may have
    initialised the GPU), relay its output, return its exit code.  N ranks need N devices unless MSDE_DP_BACKEND=gloo
    (the functional smoke mode: several gloo ranks share one GPU) -- refusing here is better than a 1-GPU number
    labelled N."""
    import subprocess
    have = torch.cuda.device_count()            # counts devices without creating a HIP context
    if have < n and os.environ.get("MSDE_DP_BACKEND") != "gloo":
        print(f"[bench] --gpus {n} but only {have} device(s) visible; set MSDE_DP_BACKEND=gloo for the one-GPU smoke "
              f"mode of the DP step structure", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    print("[bench] launching %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    try:
        for line in child.stdout:               # rank 0's JSON line (and anything else the ranks print) passes through
            sys.stdout.write(line)
            sys.stdout.flush()
        return child.wait(timeout=timeout)
    except BaseException:
        child.kill()
        raise


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch_size", type=int, default=256)
    ap.add_argument("--pool", type=int, default=4, help="distinct resident batches of the per-shape-graph mode")
    ap.add_argument("--stream", type=int, default=64,
                    help="distinct batches streamed through ONE captured graph (capacity bucket, device-built plans); "
                         "0 = the round-1 mode (one graph per resident batch)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_configs45", action="store_true", help="skip the configs[3] (sampler) / configs[4] (MD17) timings")
    ap.add_argument("--no_pipeline", action="store_true", help="skip the two-bucket mode (plans of batch t+1 built beside step t)")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from the host (no hipGraph replay)")
    ap.add_argument("--no_bf16x3", action="store_true", help="(accepted for old command lines; the bf16x3 experiment is gone)")
    ap.add_argument("--debug_dp_path", action="store_true",
                    help="one GPU: initialise a 1-rank RCCL group and run the multi-GPU step structure "
                         "(graph without Adam; all-reduce; Adam kernel)")
    ap.add_argument("--full", action="store_true",
                    help="configs[2] per-GPU work: add the 3D->2D dense head loss (default: configs[1])")
    ap.add_argument("--score_kernel", default="ops", choices=["ops", "mol"],
                    help="2D->3D score network under autograd: operator by operator (default) or one launch each way with one "
                         "workgroup per molecule (pretrain.py --score_kernel)")
    ap.add_argument("--no_config2", action="store_true",
                    help="skip the second timing (configs[2]: the full step with the 3D->2D head, same ranks) behind the headline")
    ap.add_argument("--census_only", action="store_true",
                    help="launcher check: start the ranks, count them with one all-reduce, print the line's DP keys and exit "
                         "(runs without a GPU under MSDE_DP_BACKEND=gloo: tests/test_host_logic.py)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing has touched HIP yet (torch is imported, no device
        # call made), and the ranks are CHILD processes -- this process only relays rank 0's JSON line and the exit code.
        sys.exit(spawn_ranks(a.gpus, sys.argv[1:]))

    if a.census_only:
        from moleculesde_amd import dp
        on_gpu = torch.cuda.device_count() > 0
        rank, world, local = dp.init_from_env("cuda" if on_gpu else "cpu")
        dev = torch.device("cuda", local if torch.cuda.device_count() > local else 0) if on_gpu else torch.device("cpu")
        ones = torch.ones(1, device=dev)
        mine = torch.tensor([dev.index if on_gpu else -1], device=dev, dtype=torch.int64)
        got = [mine]
        if world > 1:
            torch.distributed.all_reduce(ones)
            got = [torch.zeros_like(mine) for _ in range(world)]
            torch.distributed.all_gather(got, mine)
        if rank == 0:
            print(json.dumps({"n_gpus": world, "rccl_ranks_seen": int(ones.item()), "rank_devices": [int(g.item()) for g in got],
                              "dp_backend": torch.distributed.get_backend() if world > 1 else None, "census_only": True,
                              "dp_line_keys": list(DP_LINE_KEYS), "dp_part_keys": list(DP_PART_KEYS)}),
                  flush=True)
        dp.barrier()
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return

    from moleculesde_amd import _lib, dp, pretrain
    from moleculesde_amd.geom3d import prepare_batch
    from moleculesde_amd.synthetic import make_batch, batch_stats
    _lib.load()
    rank, world, local = dp.init_from_env("cuda", force=a.debug_dp_path)
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if world != a.gpus and rank == 0:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if os.environ.get("MSDE_DP_BACKEND") == "gloo" and torch.cuda.device_count() <= local:
        local = 0            # smoke mode: several gloo ranks share ONE GPU (DP step structure without RCCL)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    torch.manual_seed(0)

    args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=1 if a.full else 0, batch_size=a.batch_size, score_kernel=a.score_kernel)
    trainer = pretrain.Trainer(args, device)
    trainer.adam_outside_graph = a.debug_dp_path
    cpu_pool = [make_batch(a.batch_size, seed=dp.shard_seed(s, rank)) for s in range(a.pool)]
    stats = batch_stats(cpu_pool[0])
    pool = [prepare_batch(b.clone(), device) for b in cpu_pool]       # prepare_batch moves its argument to the device

    # warm-up: W eager steps (every batch shape at least once), then -- unless --eager -- each batch shape
    # is captured into a hipGraph (fwd + bwd + grad flattening + Adam) that the timed steps replay
    for s in range(max(a.warmup, len(pool))):
        trainer.step(pool[s % len(pool)])
    use_graph = not a.eager
    eager_ms = None
    if use_graph:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(len(pool)):
            trainer.step(pool[s])
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t0) / len(pool) * 1e3
        try:
            for b in pool:
                trainer.capture(b)
            for b in pool:
                trainer.step_graph(b)
        except Exception as exc:      # never lose the measurement to a capture problem: fall back to eager
            print(f"[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); running eager", file=sys.stderr)
            use_graph = False
    step_fn = trainer.step_graph if use_graph else trainer.step

    local_s = {}                # this rank's own time of the last timed() call (its device drained, BEFORE the closing barrier)

    def timed(fn, items, steps):
        import gc
        gc.collect()
        gc.disable()            # a generation-2 collection of the interpreter (tens of ms with the captured graphs'
        try:                    # autograd objects alive) inside a 30-step timing would be charged to the step
            dp.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(steps):
                fn(items[s % len(items)])
            if world > 1:       # the rank's own finishing time (the per-rank spread of the line); the barrier below then
                torch.cuda.synchronize()          # waits for the slowest rank, which is what `value` is computed from
                local_s["last"] = time.perf_counter() - t0
            dp.barrier()
            torch.cuda.synchronize()
            dt_ = time.perf_counter() - t0
            local_s.setdefault("last", dt_)
            if world == 1:
                local_s["last"] = dt_
            return dt_
        finally:
            gc.enable()

    dt_pool = timed(step_fn, pool, a.steps)
    dt, launch, stream_info = dt_pool, None, None
    dt_local = local_s.get("last", dt_pool)

    def stream_modes(tr, full, pcie_variants=True):
        """`--stream` DISTINCT batches, each fed as its raw collated arrays (one device-to-device copy of the resident blob),
        plans and the extended graph built on the device inside the ONE captured graph of trainer `tr`; then the two-bucket
        mode and (pcie_variants) the blobs from pinned host memory.  Which mode is the headline is a FIXED rule, not "the
        faster one": configs[1] -> the one-graph mode, the full step (configs[2]'s per-GPU work) -> the two-bucket mode; both
        times are reported either way.  Returns (seconds of the headline mode, this rank's own seconds, launch text, info,
        (bucket, blobs))."""
        from moleculesde_amd import bucket as BK
        t0 = time.perf_counter()
        extra = [make_batch(a.batch_size, seed=dp.shard_seed(1000 + s, rank)) for s in range(max(a.stream - len(cpu_pool), 0))]
        stream_cpu = (cpu_pool + extra)[:a.stream]
        needs = [BK.raw_sizes(b) for b in stream_cpu]
        caps = BK.Caps.covering(needs)
        bk = tr.make_bucket(caps)
        blobs = [BK.pack_raw(b, caps).to(device) for b in stream_cpu]
        host_prep_s = time.perf_counter() - t0
        tr.capture_bucket(bk, blobs[0])
        for s in range(a.warmup):
            tr.step_bucket(bk, blobs[s % len(blobs)])
        ok, _ = bk.check()
        assert ok
        dt_ = timed(lambda blob: tr.step_bucket(bk, blob), blobs, a.steps)
        dtl_ = local_s["last"]
        ok, _ = bk.check()
        assert ok
        launch_ = ("ONE hipGraph for all batches: %d distinct batches streamed as raw collated arrays (1 copy each), "
                   "plans + extend_graph built on the device inside the graph" % len(blobs))
        dt_one_graph, dt_pipe = dt_, None
        if not a.no_pipeline:
            # two buckets used alternately: the plans of batch t+1 are built (plan graph, third stream) while the
            # step of batch t runs -- the same work per batch, batch construction off the step's critical path
            pipe = pretrain.BucketPipeline(tr, caps, blobs[0])
            state = {"i": 0}
            pipe.submit(blobs[0])

            def piped_step(_):
                state["i"] += 1
                pipe.submit(blobs[state["i"] % len(blobs)])
                return pipe.step()
            for s in range(a.warmup):
                piped_step(None)
            dt_pipe = timed(piped_step, blobs, a.steps)
            dtl_pipe = local_s["last"]
            pipe.step()                       # drain the batch submitted last
            torch.cuda.synchronize()
            assert pipe.check()
            if full:
                dt_, dtl_ = dt_pipe, dtl_pipe
                launch_ = ("two capacity buckets used alternately (one captured step graph + one plan graph each): %d "
                           "distinct batches streamed as raw collated arrays (1 copy each); plans + extend_graph of "
                           "batch t+1 built on the device beside the step of batch t" % len(blobs))
        pad = {k: round(getattr(caps, k) / max(n, 1), 3) for k, n in
               (("N", sum(x["N"] for x in needs) / len(needs)), ("E_b", sum(x["E_b"] for x in needs) / len(needs)),
                ("E_e", sum(x["E_e"] for x in needs) / len(needs)), ("P", sum(x["P"] for x in needs) / len(needs)))}
        info = {"distinct_batches": len(blobs), "capacities": caps.as_dict(), "capacity_over_mean_size": pad,
                "raw_blob_bytes": int(blobs[0].numel() * 4),
                "ms_per_step_one_graph_plan_inside": round(dt_one_graph / a.steps * 1e3, 3),
                "ms_per_step_two_buckets_plan_ahead": None if dt_pipe is None else round(dt_pipe / a.steps * 1e3, 3),
                "host_prep_s_synthetic_generation_and_packing": round(host_prep_s, 2)}
        if pcie_variants:
            # PCIe-inclusive variant: the same blobs from pinned host memory
            pinned = [BK.pack_raw(b, caps, pin=True) for b in stream_cpu[:16]]
            dt_h2d = timed(lambda blob: tr.step_bucket(bk, blob), pinned, a.steps)
            # ... and prefetched one step ahead (bucket.BlobFeeder: the blob of step t+1 crosses PCIe while step t runs)
            feeder = BK.BlobFeeder(bk)
            feeder.submit(pinned[0])
            state = {"i": 0}

            def fed_step(_):
                state["i"] += 1
                feeder.submit(pinned[state["i"] % len(pinned)])
                feeder.load_next()
                return tr.step_graph(bk.batch)
            dt_fed = timed(fed_step, pinned, a.steps)
            info.update({"ms_per_step_blobs_from_pinned_host": round(dt_h2d / a.steps * 1e3, 3),
                         "ms_per_step_blobs_from_pinned_host_prefetched": round(dt_fed / a.steps * 1e3, 3)})
        return dt_, dtl_, launch_, info, (bk, blobs)

    dp_parts = None
    if a.stream > 0 and use_graph:
        try:
            dt, dt_local, launch, stream_info, (bk0, blobs0) = stream_modes(trainer, a.full)
            stream_info["ms_per_step_4_resident_batches_own_graphs"] = round(dt_pool / a.steps * 1e3, 3)
            if trainer._use_dp():
                dp_parts = dp_step_parts(trainer, lambda i: trainer.step_bucket(bk0, blobs0[i % len(blobs0)]))
        except Exception as exc:
            print(f"[bench] bucket mode failed ({type(exc).__name__}: {exc}); reporting the per-shape-graph mode",
                  file=sys.stderr)
            dt = dt_pool
        finally:
            from moleculesde_amd import hip as _hip
            _hip.clear_row_bounds()
    if dp_parts is None and trainer._use_dp() and use_graph:
        dp_parts = dp_step_parts(trainer, lambda i: trainer.step_graph(pool[i % len(pool)]))

    # configs[2] (BASELINE.json: contrastive + 2D->3D + 3D->2D, 256 molecules per GPU, DP): the SAME ranks time the full step on
    # a second Trainer right behind the headline, so that one `--gpus N` run measures both configurations
    # (examples/pretrain_MoleculeSDE.py:135-156 is the loss composition the full step adds to).  Every rank takes part:
    # the trainer's collectives and timed()'s barriers need all of them.
    config2 = None
    if not a.full and not a.no_config2 and a.stream > 0 and use_graph:
        try:
            args2 = pretrain.readme_args(SDE_coeff_generative_3Dto2D=1, batch_size=a.batch_size, score_kernel=a.score_kernel)
            tr2 = pretrain.Trainer(args2, device)
            tr2.adam_outside_graph = a.debug_dp_path
            dt2, dtl2, launch2, info2, (bk2, blobs2) = stream_modes(tr2, True, pcie_variants=False)
            parts2 = dp_step_parts(tr2, lambda i: tr2.step_bucket(bk2, blobs2[i % len(blobs2)])) if tr2._use_dp() else None
            config2 = {"dt": dt2, "dt_local": dtl2, "launch": launch2, "stream": info2, "dp_step_parts_us": parts2}
            del tr2
        except Exception as exc:
            print(f"[bench] configs[2] (full step) failed ({type(exc).__name__}: {exc})", file=sys.stderr)
            config2 = {"error": f"{type(exc).__name__}: {exc}"}
        finally:
            from moleculesde_amd import hip as _hip
            _hip.clear_row_bounds()

    ranks_seen, rank_devices, backend = 1, [int(device.index)], None
    if world > 1 or torch.distributed.is_initialized():
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
        # census: an all-reduce of ones says how many ranks really took part in the collectives, the gathered device
        # indices say whether they sat on distinct GPUs (the one-GPU gloo smoke mode shows [0, 0, ...])
        ones = torch.ones(1, device=device)
        torch.distributed.all_reduce(ones)
        ranks_seen = int(ones.item())
        mine = torch.tensor([device.index], device=device, dtype=torch.int64)
        got = [torch.zeros_like(mine) for _ in range(torch.distributed.get_world_size())]
        torch.distributed.all_gather(got, mine)
        rank_devices = [int(g.item()) for g in got]
        backend = torch.distributed.get_backend()
    # per-rank spread and the DP pieces of every rank (a slow 8-GPU number must explain itself): row r = rank r
    c2_ok = isinstance(config2, dict) and "dt" in config2
    pk = DP_PART_KEYS[:4]
    row = [dt_local, config2["dt"] if c2_ok else None, config2["dt_local"] if c2_ok else None]
    row += [(dp_parts or {}).get(k) for k in pk]
    row += [((config2 or {}).get("dp_step_parts_us") or {}).get(k) for k in pk]
    rows = gather_ranks(row, device)
    nan_none = lambda v: None if v != v else v
    if c2_ok:
        config2["dt"] = max(r[1] for r in rows)          # max over ranks, like the headline

    def spread(col, scale=1.0, nd=3):
        v = [r[col] * scale for r in rows if r[col] == r[col]]
        return None if not v else {"min": round(min(v), nd), "max": round(max(v), nd), "per_rank": [round(x, nd) for x in v]}

    def parts_out(mine, first_col):
        if not mine or "error" in mine:
            return mine
        o = dict(mine)
        o["max_over_ranks"] = {k: (spread(first_col + i, 1.0, 1) or {}).get("max") for i, k in enumerate(pk)}
        return o

    out = None
    if rank == 0:
        print(f"[bench] {a.steps} steps in {dt:.3f}s", file=sys.stderr, flush=True)
        mols = world * a.batch_size * a.steps
        # everything below is measured by rank 0 ALONE while the other ranks wait at the final barrier: the trainer must not
        # issue collectives any more (found by the 2-rank smoke run: the kernel count's eager step hung in its all-reduce)
        trainer.dp_enabled = False
        # `roofline` = the kernel with the largest share of the step's GPU time (profiles/*_kernel_stats.csv):
        # cfconv_fused_bwd_w, timed at the workgroup count the step launches it with; the full-width figures are
        # kept as separate keys
        roof_filter = roofline_pair_filter(trainer, pool[0])
        roof_bwd = roofline_pair_bwd_w(trainer, pool[0])
        in_step = in_step_stamps(trainer, cpu_pool[0].clone(), device)
        roof_agg = roofline_hbm_kernel(trainer, pool[0], in_step)
        family = in_step_family(trainer, cpu_pool[0].clone(), device)
        roof = roofline_gemm(trainer, cpu_pool[0].clone(), device, in_step, family)
        n_kern, n_foreign = count_kernels_per_step(trainer, pool[0])
        roof_head = roofline_dense_head_node_mlp(pool[0])
        roof_forward = roofline_forward(trainer, pool[0], stats)
        out = {
            "metric": "molecules/sec pretrain step (SchNet+SDE VE, bs256)",
            "value": round(mols / dt, 1), "unit": "molecules/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "rccl_ranks_seen": ranks_seen, "rank_devices": rank_devices, "dp_backend": backend,
            "ms_per_step_per_rank": spread(0, 1e3 / a.steps),
            "dp_step_parts_us": parts_out(dp_parts, 3),
            "config2_full": None,
            "config": {"workload": "PCQM4Mv2-shaped pretrain step: GIN5x300 + SchNet(6x128f,51g,rc10) + "
                                   "EBM_node_dot_prod contrastive + SDEModel2Dto3D_02 VE" + (" + SDEModel3Dto2D_node_adj_dense VE" if a.full else "") + "; fwd+bwd+Adam",
                       "molecules_per_gpu": a.batch_size, "global_batch": world * a.batch_size,
                       "parallelism": f"dp{world}", "batch_shape": stats, "dropout_p_2Dto3D": 0.1,
                       "launch": launch or (("hipGraph replay, one graph per batch shape (pool of %d shapes)" % len(pool))
                                            if use_graph else "eager"),
                       "stream": stream_info, "eager_ms_per_step": None if eager_ms is None else round(eager_ms, 3),
                       "loss_scalar": float(trainer.log["2Dto3D"]) / max(trainer.steps, 1)},
            "tail_us": None if in_step.get("tail") is None else round(in_step["tail"], 1),
            "tail_us_what": "end of the backward chain of the main stream -> end of Adam and of the weight-copy refresh, device "
                            "stamps inside the captured per-shape step (grouped weight gradients, slab reduction, Adam)",
            "tail_after_both_streams_us": None if in_step.get("tail_both") is None else round(in_step["tail_both"], 1),
            "tail_after_both_streams_what": "end of the SECOND stream's work of the backward pass INCLUDING its deferred leaf kernels "
                                            "(embedding / bond-table gradients, which run beside the grouped weight-gradient launch) -> "
                                            "end of Adam and of the weight-copy refresh",
            "tail_after_side_chain_us": None if in_step.get("tail_chain") is None else round(in_step["tail_chain"], 1),
            "tail_after_side_chain_what": "end of the second stream's backward CHAIN (SchNet's backward, before its leaf kernels) -> "
                                          "end of the step: with tail_us, how serial the grouped launch + reduction + Adam are",
            "step_us_device_stamps": None if in_step.get("step") is None else round(in_step["step"], 1),
            "kernels_per_step": n_kern, "kernels_per_step_not_from_libmsde_hip": n_foreign,
            "roofline": roof,
            "roofline_cfconv_pair_filter": roof_filter,
            "roofline_cfconv_pair_bwd_w": roof_bwd,
            "roofline_hbm_message_passing": roof_agg,
            "roofline_forward_schnet_sde2d3d": roof_forward,
            "roofline_dense_head_node_mlp": roof_head,
        }
        if c2_ok:
            out["config2_full"] = {
                "what": "BASELINE.json configs[2]: the FULL pretrain step (contrastive + 2D->3D + 3D->2D dense head, VE; "
                        "examples/pretrain_MoleculeSDE.py:135-156), %d molecules per GPU, timed by the same %d rank(s) right behind "
                        "the headline with the same barrier + synchronize bracket, max over ranks" % (a.batch_size, world),
                "ms_per_step": round(config2["dt"] / a.steps * 1e3, 3),
                "value": round(world * a.batch_size * a.steps / config2["dt"], 1), "unit": "molecules/s", "n_gpus": world,
                "steps": a.steps, "launch": config2["launch"], "stream": config2["stream"],
                "ms_per_step_per_rank": spread(2, 1e3 / a.steps),
                "dp_step_parts_us": parts_out(config2.get("dp_step_parts_us"), 7)}
        elif a.full:
            out["config2_full"] = {"what": "this line's headline IS the full step (--full)", "ms_per_step": out["ms_per_step"],
                                   "value": out["value"], "unit": "molecules/s", "n_gpus": world}
        elif config2 is not None:
            out["config2_full"] = config2              # {"error": ...}
        assert all(k in out for k in DP_LINE_KEYS)
        if world == 1 and not a.no_configs45:
            out["config4_sampler"] = config4_sampler(device)
            out["config5_md17"] = config5_md17(device)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.batch_size)
        print(json.dumps(out), flush=True)
    dp.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
