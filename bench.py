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

Rank 0 prints ONE JSON line; `roofline` = the kernel with the largest share of the step's GPU time, timed live with
HIP events at the step's launch geometry; `roofline_forward_schnet_sde2d3d` = the north-star forward figure;
`cpu_baseline` = the oracle port timed on the host cores (N=1 only).
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


def _radius(trainer, batch):
    from moleculesde_amd import hip, plan as P
    sch = trainer.models["model_3D"]
    pl = P.get_plan(batch)
    with torch.no_grad():
        rplan, dist = hip.radius_plan(batch.positions, pl.batch_i32, pl.mol_ptr, sch.cutoff, pl.E_r_cap, 32)
    return sch, rplan, dist, int(rplan.rowptr[-1]), batch.x.size(0)


def _pmc_traffic(kernel, E, N, wgs=None):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes (profiles/r02_pmc_counters.json: separate
    FETCH_SIZE / WRITE_SIZE passes of tools/prof_kernels.py, corrected as MI355X_MICROARCH.md prescribes), valid only for
    the batch shape they were collected on; `wgs` selects the launch width (grid = workgroups x 256 threads)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r02_pmc_counters.json")) as f:
            d = json.load(f)
        if d.get("shape", {}).get("E_r") != E or d.get("shape", {}).get("N") != N:
            return None
        cands = {k: v for k, v in d.items() if k.startswith(kernel + "[grid=") and "traffic_bytes" in v}
        if not cands:
            return None
        if wgs:
            key = f"{kernel}[grid={int(wgs) * 256}]"
            return cands[key]["traffic_bytes"] if key in cands else None
        return cands[max(cands, key=lambda k: int(k.split("=")[1].rstrip("]")))]["traffic_bytes"]
    except Exception:
        return None


def _rocprof_avg_us(kernel, full=False):
    """Average duration of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of this same command
    (profiles/r02_{default,full}_bench_kernel_stats.csv): launches inside the step (beside the other stream's kernels)
    and the roofline loop's own, together.  None when the summary does not hold the kernel."""
    import csv
    try:
        path = os.path.join(ROOT, "profiles", "r02_%s_bench_kernel_stats.csv" % ("full" if full else "default"))
        with open(path) as f:
            for r in csv.DictReader(f):
                if kernel in r["Name"]:
                    return round(float(r["AverageNs"]) / 1e3, 2)
    except Exception:
        pass
    return None


def _step_widths(trainer):
    """Workgroup counts the trainer gives the two wide CFConv kernels inside the step (pretrain.Trainer.losses:
    narrowed while SchNet runs beside the GIN -> 2D->3D chain, full width with the 3D->2D head behind it)."""
    from moleculesde_amd import pretrain
    side = trainer.overlap_streams and not (trainer.args.SDE_coeff_generative_3Dto2D > 0)
    return (pretrain.SIDE_CFCONV_FWD_WGS, pretrain.SIDE_CFCONV_BWD_WGS) if side else (None, None)


def roofline_fused_fwd(trainer, batch, iters=50, wgs=None):
    """The fused CFConv forward (csrc/cfconv_fused.hip; 6 launches per step forward), fp32 matrix-core bound
    (83 FLOP/B): algorithmic FLOPs per launch = E * 2 * (G*F + F*F) for the filter network (SURVEY §8d row 'SchNet
    CFConv fused'); algorithmic bytes = E*8 + E*F*4 (gather) + weights + N*F*4 (+ E*F*4 filter rows out).
    Timed on the raw C-ABI call with HIP events at the SAME workgroup count the step launches it with (`wgs`; None
    = full width), so the figure is the kernel as the step runs it, minus cross-stream contention."""
    from moleculesde_amd import hip, _lib
    sch, rplan, dist, E, N = _radius(trainer, batch)
    if sch.num_filters != 128:
        return None
    blk, de = sch.interactions[0], sch.distance_expansion
    G = sch.num_gaussians
    with torch.no_grad():
        x1 = torch.randn(N, 128, device=batch.x.device)
        W1T = blk.mlp[0].weight.detach()          # nn.Linear layouts, as the C ABI takes them
        W2T = blk.mlp[2].weight.detach()
        b1, b2 = blk.mlp[0].bias.detach(), blk.mlp[2].bias.detach()
        agg = torch.empty(N, 128, device=batch.x.device)
        Wf = torch.empty(rplan.E, 128, device=batch.x.device)
        stream = torch.cuda.current_stream()
        p, st = hip._p, hip._stream()
        cpw = hip.FUSED_CHUNKS_PER_WG if wgs is None else max(1, -(-((rplan.E + 31) // 32) // int(wgs)))
        fn = lambda: _lib.call("msde_cfconv_fused_fwd", p(x1), p(dist), p(rplan.rowptr), p(rplan.src), p(rplan.dst), p(W1T),
                               p(b1), p(W2T), p(b2), p(de.offset), N, 128, G, rplan.E, float(de.coeff), float(sch.cutoff),
                               cpw, p(agg), p(Wf), st)
        ms = _event_time_ms(fn, iters, stream)
    flops = E * 2.0 * (G * 128 + 128 * 128)
    nbytes = E * 8 + E * 128 * 4 * 2 + (G * 128 + 128 * 128 + 256) * 4 + N * 128 * 4 + (N + 1) * 4
    tf = flops / (ms * 1e-3) / 1e12
    return {"kernel": "cfconv_fused_fwd_kernel", "bound": "mfma", "achieved": round(tf, 2), "peak": FP32_MFMA_PEAK_TF,
            "unit": "TFLOP/s", "frac": round(tf / FP32_MFMA_PEAK_TF, 4),
            "traffic": _pmc_traffic("cfconv_fused_fwd_kernel", E, N, wgs), "flops_per_launch": flops,
            "algorithmic_bytes_per_launch": nbytes, "hbm_frac_at_this_time": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "avg_launch_us": round(ms * 1e3, 2), "launches_per_step": 6, "edges": E, "nodes": N,
            "workgroups": "full width" if wgs is None else int(wgs)}


def roofline_fused_bwd(trainer, batch, iters=30, wgs=None):
    """Second line: the recomputing weight-gradient kernel of the fused CFConv (cfconv_fused_bwd.hip),
    FLOPs per launch = E * 2 * (G*F [recompute pre1] + F*F [g_W2] + F*F [W2^T g] + F*G [g_W1])."""
    from moleculesde_amd import hip, _lib
    sch, rplan, dist, E, N = _radius(trainer, batch)
    if sch.num_filters != 128:
        return None
    blk, de = sch.interactions[0], sch.distance_expansion
    G = sch.num_gaussians
    with torch.no_grad():
        dev = batch.x.device
        x1 = torch.randn(N, 128, device=dev)
        g = torch.randn(N, 128, device=dev)
        W1, b1, W2 = blk.mlp[0].weight.detach(), blk.mlp[0].bias.detach(), blk.mlp[2].weight.detach()
        gW1, gb1, gW2, gb2 = torch.empty_like(W1), torch.empty_like(b1), torch.empty_like(W2), torch.empty_like(b1)
        mw = int(wgs or 0)
        ws = hip._cf_workspace(rplan.E, G, dev, mw)
        p, st = hip._p, hip._stream()
        fn = lambda: _lib.call("msde_cfconv_fused_bwd_w", p(g), p(x1), p(dist), p(rplan.rowptr), p(rplan.src), p(rplan.dst),
                               p(W1), p(b1), p(W2), p(de.offset), N, 128, G, rplan.E, float(de.coeff), float(sch.cutoff),
                               mw, p(None), p(None), p(None), p(None), p(ws), st)      # slabs only: the kernel alone
        ms = _event_time_ms(fn, iters, torch.cuda.current_stream())
    flops = E * 2.0 * (2 * G * 128 + 2 * 128 * 128)
    tf = flops / (ms * 1e-3) / 1e12
    full = trainer.args.SDE_coeff_generative_3Dto2D > 0
    rp = _rocprof_avg_us("cfconv_fused_bwd_w_pipe_kernel", full)
    return {"kernel": "cfconv_fused_bwd_w_pipe_kernel", "bound": "mfma", "achieved": round(tf, 2),
            "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / FP32_MFMA_PEAK_TF, 4),
            "rocprofv3_avg_launch_us_same_command": rp,
            "frac_at_rocprofv3_avg": None if not rp else round(flops / (rp * 1e-6) / 1e12 / FP32_MFMA_PEAK_TF, 4),
            "traffic": _pmc_traffic("cfconv_fused_bwd_w_pipe_kernel", E, N, wgs), "flops_per_launch": flops,
            "avg_launch_us": round(ms * 1e3, 2), "launches_per_step": 6, "edges": E, "nodes": N,
            "workgroups": "full width" if wgs is None else int(wgs)}


def roofline_hbm_kernel(trainer, batch, iters=50):
    """HBM-bound message-passing kernel that the step really launches (6 x per backward pass): the input gradient
    of the CFConv aggregation, g_x1[j] = sum_{e: src_e = j} g_agg[dst_e] * Wf[e]  (schnet.py:190,194-195 transposed;
    `cfconv_aggregate_bwd_x_kernel`).  Algorithmic bytes per launch (SURVEY §8d convention): E*F*4 (filter rows) +
    E*F*4 (gathered gradient rows) + E*8 (slot -> edge, edge -> target) + (N+1)*4 + N*F*4 (output)."""
    from moleculesde_amd import hip, _lib
    sch, rplan, dist, E, N = _radius(trainer, batch)
    Fd = sch.num_filters
    with torch.no_grad():
        dev = batch.x.device
        g = torch.randn(N, Fd, device=dev)
        Wf = torch.randn(rplan.E, Fd, device=dev)
        out = torch.empty(N, Fd, device=dev)
        p, st = hip._p, hip._stream()
        fn = lambda: _lib.call("msde_cfconv_aggregate_bwd_x", p(g), p(Wf), p(None), p(rplan.rowptr_s), p(rplan.perm_s),
                               p(rplan.dst), N, Fd, p(out), st)
        ms = _event_time_ms(fn, iters, torch.cuda.current_stream())
    nbytes = E * Fd * 4 * 2 + E * 8 + (N + 1) * 4 + N * Fd * 4
    achieved = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": "cfconv_aggregate_bwd_x_kernel", "bound": "hbm", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": _pmc_traffic("cfconv_aggregate_bwd_x_kernel", E, N), "bytes_per_launch": nbytes,
            "avg_launch_us": round(ms * 1e3, 2), "launches_per_step": 6}


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

    def fwd():
        _, h3 = sch(batch.x[:, 0], batch.positions, batch.batch, return_latent=True)
        l23 = sde(h2, batch, anneal_power=0)["position"]
        keep[:] = [h3, l23]

    for _ in range(2):
        fwd()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
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
    B = stats["B"]
    t = ms * 1e-3
    gbs, tf = nbytes / t / 1e9, flops / t / 1e12
    return {"what": "SchNet + SDEModel2Dto3D_02 forward, training mode, bs %d (%s)" % (B, how), "ms": round(ms, 4),
            "molecules_per_s_forward_only": round(B / t, 1),
            "algorithmic_MB": round(nbytes / 1e6, 2), "algorithmic_MB_per_molecule": round(nbytes / 1e6 / B, 4),
            "algorithmic_GFLOP": round(flops / 1e9, 3), "algorithmic_MFLOP_per_molecule": round(flops / 1e6 / B, 2),
            "hbm": {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)},
            "fp32_flop_floor": {"achieved": round(tf, 2), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                "frac": round(tf / FP32_MFMA_PEAK_TF, 4)},
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

        def chain():
            hip.gemm_ex(X, W[0], F1, bias=b[0], act="silu", Z=Z1)
            hip.gemm_ex(F1, W[1], F2, bias=b[1], act="silu", Z=Z2)
            hip.gemm_ex(F2, W[2], OUT[:, :119], bias=b[2])
        for _ in range(3):
            chain()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            chain()
        ms = _event_time_ms(g.replay, iters, torch.cuda.current_stream())
    flops = 2.0 * N * (364 * 728 + 728 * 728 + 728 * 119)
    tf = flops / (ms * 1e-3) / 1e12
    return {"kernel": "gemm_ex_kernel x3 (node MLP 364->728->728->119 of the dense head, valid atoms only)", "bound": "mfma",
            "achieved": round(tf, 2), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / FP32_MFMA_PEAK_TF, 4),
            "us_per_chain": round(ms * 1e3, 2), "rows": N, "flops": flops}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch_size", type=int, default=256)
    ap.add_argument("--pool", type=int, default=4, help="distinct resident batches of the per-shape-graph mode")
    ap.add_argument("--stream", type=int, default=64,
                    help="distinct batches streamed through ONE captured graph (capacity bucket, device-built plans); "
                         "0 = the round-1 mode (one graph per resident batch)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from the host (no hipGraph replay)")
    ap.add_argument("--debug_dp_path", action="store_true",
                    help="one GPU: initialise a 1-rank RCCL group and run the multi-GPU step structure "
                         "(graph without Adam; all-reduce; Adam kernel)")
    ap.add_argument("--full", action="store_true",
                    help="configs[2] per-GPU work: add the 3D->2D dense head loss (default: configs[1])")
    a = ap.parse_args()

    from moleculesde_amd import _lib, dp, pretrain
    from moleculesde_amd.geom3d import prepare_batch
    from moleculesde_amd.synthetic import make_batch, batch_stats
    _lib.load()
    rank, world, local = dp.init_from_env("cuda", force=a.debug_dp_path)
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if world != a.gpus and rank == 0:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    torch.manual_seed(0)

    args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=1 if a.full else 0, batch_size=a.batch_size)
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

    def timed(fn, items, steps):
        dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(steps):
            fn(items[s % len(items)])
        dp.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    dt_pool = timed(step_fn, pool, a.steps)
    dt, launch, stream_info = dt_pool, None, None
    if a.stream > 0 and use_graph:
        # headline mode: `--stream` DISTINCT batches, each fed as its raw collated arrays (one device-to-device copy of
        # the resident blob), plans and the extended graph built on the device inside the ONE captured graph
        from moleculesde_amd import bucket as BK
        t0 = time.perf_counter()
        extra = [make_batch(a.batch_size, seed=dp.shard_seed(1000 + s, rank)) for s in range(max(a.stream - len(cpu_pool), 0))]
        stream_cpu = (cpu_pool + extra)[:a.stream]
        needs = [BK.raw_sizes(b) for b in stream_cpu]
        caps = BK.Caps.covering(needs)
        bk = trainer.make_bucket(caps)
        blobs = [BK.pack_raw(b, caps).to(device) for b in stream_cpu]
        host_prep_s = time.perf_counter() - t0
        try:
            trainer.capture_bucket(bk, blobs[0])
            for s in range(a.warmup):
                trainer.step_bucket(bk, blobs[s % len(blobs)])
            ok, _ = bk.check()
            assert ok
            dt = timed(lambda blob: trainer.step_bucket(bk, blob), blobs, a.steps)
            ok, _ = bk.check()
            assert ok
            launch = ("ONE hipGraph for all batches: %d distinct batches streamed as raw collated arrays (1 copy each), "
                      "plans + extend_graph built on the device inside the graph" % len(blobs))
            pad = {k: round(getattr(caps, k) / max(n, 1), 3) for k, n in
                   (("N", sum(x["N"] for x in needs) / len(needs)), ("E_b", sum(x["E_b"] for x in needs) / len(needs)),
                    ("E_e", sum(x["E_e"] for x in needs) / len(needs)), ("P", sum(x["P"] for x in needs) / len(needs)))}
            # PCIe-inclusive variant: the same blobs from pinned host memory
            pinned = [BK.pack_raw(b, caps, pin=True) for b in stream_cpu[:16]]
            dt_h2d = timed(lambda blob: trainer.step_bucket(bk, blob), pinned, a.steps)
            # ... and prefetched one step ahead (bucket.BlobFeeder: the blob of step t+1 crosses PCIe while step t runs)
            feeder = BK.BlobFeeder(bk)
            feeder.submit(pinned[0])
            state = {"i": 0}

            def fed_step(_):
                state["i"] += 1
                feeder.submit(pinned[state["i"] % len(pinned)])
                feeder.load_next()
                return trainer.step_graph(bk.batch)
            dt_fed = timed(fed_step, pinned, a.steps)
            stream_info = {"distinct_batches": len(blobs), "capacities": caps.as_dict(), "capacity_over_mean_size": pad,
                           "raw_blob_bytes": int(blobs[0].numel() * 4),
                           "ms_per_step_blobs_from_pinned_host": round(dt_h2d / a.steps * 1e3, 3),
                           "ms_per_step_blobs_from_pinned_host_prefetched": round(dt_fed / a.steps * 1e3, 3),
                           "ms_per_step_4_resident_batches_own_graphs": round(dt_pool / a.steps * 1e3, 3),
                           "host_prep_s_synthetic_generation_and_packing": round(host_prep_s, 2)}
        except Exception as exc:
            print(f"[bench] bucket mode failed ({type(exc).__name__}: {exc}); reporting the per-shape-graph mode",
                  file=sys.stderr)
            dt = dt_pool
        finally:
            from moleculesde_amd import hip as _hip
            _hip.clear_row_bounds()
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)

    out = None
    if rank == 0:
        print(f"[bench] {a.steps} steps in {dt:.3f}s", file=sys.stderr, flush=True)
        mols = world * a.batch_size * a.steps
        # `roofline` = the kernel with the largest share of the step's GPU time (profiles/*_kernel_stats.csv):
        # cfconv_fused_bwd_w, timed at the workgroup count the step launches it with; the full-width figures are
        # kept as separate keys
        wf, wb = _step_widths(trainer)
        roof = roofline_fused_bwd(trainer, pool[0], wgs=wb)
        roof["standalone_full_width"] = roofline_fused_bwd(trainer, pool[0]) if wb else None
        roof_fwd = roofline_fused_fwd(trainer, pool[0], wgs=wf)
        roof_fwd["standalone_full_width"] = roofline_fused_fwd(trainer, pool[0]) if wf else None
        roof_agg = roofline_hbm_kernel(trainer, pool[0])
        roof_head = roofline_dense_head_node_mlp(pool[0])
        roof_forward = roofline_forward(trainer, pool[0], stats)
        out = {
            "metric": "molecules/sec pretrain step (SchNet+SDE VE, bs256)",
            "value": round(mols / dt, 1), "unit": "molecules/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "PCQM4Mv2-shaped pretrain step: GIN5x300 + SchNet(6x128f,51g,rc10) + "
                                   "EBM_node_dot_prod contrastive + SDEModel2Dto3D_02 VE" + (" + SDEModel3Dto2D_node_adj_dense VE" if a.full else "") + "; fwd+bwd+Adam",
                       "molecules_per_gpu": a.batch_size, "global_batch": world * a.batch_size,
                       "parallelism": f"dp{world}", "batch_shape": stats, "dropout_p_2Dto3D": 0.1,
                       "launch": launch or (("hipGraph replay, one graph per batch shape (pool of %d shapes)" % len(pool))
                                            if use_graph else "eager"),
                       "stream": stream_info, "eager_ms_per_step": None if eager_ms is None else round(eager_ms, 3),
                       "loss_scalar": float(trainer.log["2Dto3D"]) / max(trainer.steps, 1)},
            "roofline": roof,
            "roofline_cfconv_fused_fwd": roof_fwd,
            "roofline_hbm_message_passing": roof_agg,
            "roofline_forward_schnet_sde2d3d": roof_forward,
            "roofline_dense_head_node_mlp": roof_head,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.batch_size)
        print(json.dumps(out), flush=True)
    dp.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
