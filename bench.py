"""bench.py — molecules/s of one MoleculeSDE pretrain step on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): PCQM4Mv2-shaped synthetic batches, 256 molecules per GPU,
GIN(5x300) + SchNet(6 interactions, 128 filters, 51 Gaussians, cutoff 10) + contrastive
(EBM_node_dot_prod, so SchNet receives gradient) + SDEModel2Dto3D_02 VE; one "step" = forward,
backward, (all-reduce), Adam, inputs already resident in HBM.  Weak scaling: every rank owns its own
shard of molecules; the only collective is one RCCL all-reduce of the flat 3.5 M-element gradient.

Rank 0 prints ONE JSON line; `roofline` is measured live with HIP events on the launch stream for the
dominant hand-written kernel, `cpu_baseline` is the oracle port timed on the host cores (N=1 only).
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


def _pmc_traffic(kernel, E, N):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes (profiles/r01_pmc_traffic.json),
    valid only for the batch shape they were collected on."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            d = json.load(f)
        ent = d.get(kernel)
        if ent and d.get("shape", {}).get("E_r") == E and d.get("shape", {}).get("N") == N:
            return ent["traffic_bytes"]
    except Exception:
        pass
    return None


def roofline_fused_fwd(trainer, batch, iters=50):
    """Roofline of the dominant hand-written kernel of the step: the fused CFConv forward
    (csrc/cfconv_fused.hip; 6 launches per step forward).  It is matrix-core bound in fp32 (83 FLOP/B):
    algorithmic FLOPs per launch = E * 2 * (G*F + F*F) for the filter network (SURVEY §8d row 'SchNet
    CFConv fused'); algorithmic bytes = E*8 + E*F*4 (gather) + weights + N*F*4 (+ E*F*4 filter rows out).
    Timed on the raw C-ABI call (weights pre-transposed, output zeroing included) with HIP events."""
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
        fn = lambda: _lib.call("msde_cfconv_fused_fwd", p(x1), p(dist), p(rplan.rowptr), p(rplan.src), p(rplan.dst), p(W1T),
                               p(b1), p(W2T), p(b2), p(de.offset), N, 128, G, rplan.E, float(de.coeff), float(sch.cutoff),
                               hip.FUSED_CHUNKS_PER_WG, p(agg), p(Wf), st)
        ms = _event_time_ms(fn, iters, stream)
    flops = E * 2.0 * (G * 128 + 128 * 128)
    nbytes = E * 8 + E * 128 * 4 * 2 + (G * 128 + 128 * 128 + 256) * 4 + N * 128 * 4 + (N + 1) * 4
    tf = flops / (ms * 1e-3) / 1e12
    return {"kernel": "cfconv_fused_fwd_kernel", "bound": "mfma", "achieved": round(tf, 2), "peak": FP32_MFMA_PEAK_TF,
            "unit": "TFLOP/s", "frac": round(tf / FP32_MFMA_PEAK_TF, 4),
            "traffic": _pmc_traffic("cfconv_fused_fwd_kernel", E, N), "flops_per_launch": flops,
            "algorithmic_bytes_per_launch": nbytes, "hbm_frac_at_this_time": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "avg_launch_us": round(ms * 1e3, 2), "launches_per_step": 6, "edges": E, "nodes": N}


def roofline_fused_bwd(trainer, batch, iters=30):
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
        ws = hip._cf_workspace(rplan.E, G, dev)
        p, st = hip._p, hip._stream()
        fn = lambda: _lib.call("msde_cfconv_fused_bwd_w", p(g), p(x1), p(dist), p(rplan.rowptr), p(rplan.src), p(rplan.dst),
                               p(W1), p(b1), p(W2), p(de.offset), N, 128, G, rplan.E, float(de.coeff), float(sch.cutoff),
                               0, p(gW1), p(gb1), p(gW2), p(gb2), p(ws), st)
        ms = _event_time_ms(fn, iters, torch.cuda.current_stream())
    flops = E * 2.0 * (2 * G * 128 + 2 * 128 * 128)
    tf = flops / (ms * 1e-3) / 1e12
    return {"kernel": "cfconv_fused_bwd_w_kernel (+ slab reduce)", "bound": "mfma", "achieved": round(tf, 2),
            "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / FP32_MFMA_PEAK_TF, 4),
            "avg_launch_us": round(ms * 1e3, 2), "launches_per_step": 6}


def roofline_cfconv_aggregate(trainer, batch, iters=50):
    """HBM-bound message-passing kernel of the decomposed path (gather x1[src] * filter, segmented sum;
    schnet.py:190,194-195).  Algorithmic bytes per launch (SURVEY §8d convention):
    E*F*4 (filter rows) + E*F*4 (gathered x1 rows) + E*8 + (N+1)*4 + N*F*4 (output)."""
    from moleculesde_amd import hip
    sch, rplan, dist, E, N = _radius(trainer, batch)
    Fd = sch.num_filters
    with torch.no_grad():
        x1 = torch.randn(N, Fd, device=batch.x.device)
        Wf = torch.randn(rplan.E, Fd, device=batch.x.device)
        C = torch.rand(rplan.E, device=batch.x.device)
        ms = _event_time_ms(lambda: hip.cfconv_aggregate(x1, Wf, C, rplan), iters, torch.cuda.current_stream())
    nbytes = E * Fd * 4 * 2 + E * 8 + (N + 1) * 4 + N * Fd * 4
    achieved = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": "cfconv_aggregate_fwd_kernel", "bound": "hbm", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": _pmc_traffic("cfconv_aggregate_fwd_kernel", E, N), "bytes_per_launch": nbytes,
            "avg_launch_us": round(ms * 1e3, 2)}


def roofline_dense_head_gemm(batch, iters=30):
    """Dense node-adjacency score head (SDEModel3Dto2D_node_adj_dense, --full): its largest product is the node
    head's Linear(728, 728) over the B*N_max padded atom slots.  Timed: the hand-written fp32-MFMA GEMM
    (msde_linear_fwd, csrc/linear.hip) on that shape, against the fp32 matrix-core peak."""
    from moleculesde_amd import hip, _lib
    dev = batch.x.device
    pl = getattr(batch, "_msde_plan", None)
    M = int(pl.B * pl.N_max) if pl is not None else 5120
    N = K = 728
    with torch.no_grad():
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev)
        b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        p, st = hip._p, hip._stream()
        fn = lambda: _lib.call("msde_linear_fwd", p(x), p(w), p(b), M, N, K, p(y), st)
        ms = _event_time_ms(fn, iters, torch.cuda.current_stream())
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    return {"kernel": "gemm_f32_mfma_kernel (Linear 728x728 of the dense head)", "bound": "mfma", "achieved": round(tf, 2),
            "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / FP32_MFMA_PEAK_TF, 4),
            "avg_launch_us": round(ms * 1e3, 2), "shape": [M, N, K]}


def cpu_baseline(bs=256, warm=1, timed=3):
    """The oracle port of the same step (oracle/restate.py, plain PyTorch fp32 on the host cores) on a
    bounded sample: `timed` steps of one bs-256 synthetic batch after `warm` warm-up steps."""
    from oracle import restate as R
    from moleculesde_amd.synthetic import make_batch
    # thousands of small eager ops per step: beyond ~16 threads the per-op barrier dominates, so the
    # port is timed on min(host cores, 16) threads (the count actually used is what `cores` reports)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    models = R.build_models(use_3d2d=False)
    opt = R.make_optimizer(models, lr=1e-4, gnn_2d_lr_scale=1.0, gnn_3d_lr_scale=0.1)
    b = make_batch(bs, seed=0)
    times = []
    for i in range(warm + timed):
        t0 = time.perf_counter()
        loss, _ = R.pretrain_losses(models, b, T=0.1, coeff_3d2d=0.0)
        opt.zero_grad()
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        print(f"[cpu_baseline] step {i}: {dt:.2f}s", file=sys.stderr, flush=True)
        if i >= warm:
            times.append(dt)
        if dt > 15.0 and times:      # keep the default run bounded on slow hosts
            break
    med = sorted(times)[len(times) // 2]
    return {"value": round(bs / med, 1), "unit": "molecules/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} timed steps (median) of one bs-{bs} synthetic batch after {warm} warm-up, "
                      f"oracle/restate.py on torch CPU fp32, {cores} threads", "ms_per_step": round(med * 1e3, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch_size", type=int, default=256)
    ap.add_argument("--pool", type=int, default=4, help="distinct synthetic batches cycled through")
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
    pool = [prepare_batch(b, device) for b in cpu_pool]

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
    dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(a.steps):
        step_fn(pool[s % len(pool)])
    dp.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)

    out = None
    if rank == 0:
        print(f"[bench] {a.steps} steps in {dt:.3f}s", file=sys.stderr, flush=True)
        mols = world * a.batch_size * a.steps
        roof = roofline_fused_fwd(trainer, pool[0])
        roof_bwd = roofline_fused_bwd(trainer, pool[0])
        roof_agg = roofline_cfconv_aggregate(trainer, pool[0])
        roof_head = roofline_dense_head_gemm(pool[0])
        out = {
            "metric": "molecules/sec pretrain step (SchNet+SDE VE, bs256)",
            "value": round(mols / dt, 1), "unit": "molecules/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "PCQM4Mv2-shaped pretrain step: GIN5x300 + SchNet(6x128f,51g,rc10) + "
                                   "EBM_node_dot_prod contrastive + SDEModel2Dto3D_02 VE" + (" + SDEModel3Dto2D_node_adj_dense VE" if a.full else "") + "; fwd+bwd+Adam",
                       "molecules_per_gpu": a.batch_size, "global_batch": world * a.batch_size,
                       "parallelism": f"dp{world}", "batch_shape": stats, "dropout_p_2Dto3D": 0.1,
                       "launch": ("hipGraph replay, one graph per batch shape (pool of %d shapes)" % len(pool))
                       if use_graph else "eager", "eager_ms_per_step": None if eager_ms is None else round(eager_ms, 3),
                       "loss_scalar": float(trainer.log["2Dto3D"]) / max(trainer.steps, 1)},
            "roofline": roof,
            "roofline_cfconv_fused_bwd_w": roof_bwd,
            "roofline_cfconv_aggregate_hbm": roof_agg,
            "roofline_dense_head_gemm": roof_head,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.batch_size)
        print(json.dumps(out), flush=True)
    dp.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
