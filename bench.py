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


def roofline_cfconv(trainer, batch, iters=50):
    """Roofline of the dominant hand-written kernel of the step: the CFConv message-passing kernel
    (gather x1[src] * filter, segmented sum per target; schnet.py:190,194-195), one launch per
    interaction.  Algorithmic bytes per launch (SURVEY §8d convention: each distinct input once, a
    gather counts E*row_bytes):  E*F*4 (filter rows) + E*F*4 (gathered x1 rows) + E*4 (cutoff) +
    E*4 (src) + (N+1)*4 (rowptr) + N*F*4 (output)."""
    from moleculesde_amd import hip, plan as P
    sch = trainer.models["model_3D"]
    pl = P.get_plan(batch)
    with torch.no_grad():
        rplan, dist = hip.radius_plan(batch.positions, pl.batch_i32, pl.mol_ptr, sch.cutoff, pl.E_r_cap, 32)
        E = int(rplan.rowptr[-1])
        N, Fd = batch.x.size(0), sch.num_filters
        x1 = torch.randn(N, Fd, device=batch.x.device)
        Wf = torch.randn(rplan.E, Fd, device=batch.x.device)
        C = torch.rand(rplan.E, device=batch.x.device)
        stream = torch.cuda.current_stream()
        ms = _event_time_ms(lambda: hip.cfconv_aggregate(x1, Wf, C, rplan), iters, stream)
    nbytes = E * Fd * 4 * 2 + E * 8 + (N + 1) * 4 + N * Fd * 4
    achieved = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": "cfconv_aggregate_fwd_kernel", "bound": "hbm", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "bytes_per_launch": nbytes, "avg_launch_us": round(ms * 1e3, 2), "launches_per_step": 6,
            "edges": E, "nodes": N}


def roofline_fused(trainer, batch, iters=50):
    """Secondary line: the fused fp32-MFMA CFConv forward (inference path).  FLOPs per launch:
    E * 2 * (G*F + F*F) (filter MLP) ; bound by the fp32 matrix peak."""
    from moleculesde_amd import hip, plan as P
    sch = trainer.models["model_3D"]
    if sch.num_filters != 128:
        return None
    pl = P.get_plan(batch)
    blk = sch.interactions[0]
    de = sch.distance_expansion
    with torch.no_grad():
        rplan, dist = hip.radius_plan(batch.positions, pl.batch_i32, pl.mol_ptr, sch.cutoff, pl.E_r_cap, 32)
        E = int(rplan.rowptr[-1])
        N = batch.x.size(0)
        x1 = torch.randn(N, 128, device=batch.x.device)
        stream = torch.cuda.current_stream()
        fn = lambda: hip.cfconv_fused_forward(x1, dist, rplan, blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight,
                                              blk.mlp[2].bias, de.offset, de.coeff, sch.cutoff)
        ms = _event_time_ms(fn, iters, stream)
    G = sch.num_gaussians
    flops = E * 2.0 * (G * 128 + 128 * 128)
    tf = flops / (ms * 1e-3) / 1e12
    return {"kernel": "cfconv_fused_fwd_kernel", "bound": "mfma", "achieved": round(tf, 2), "peak": FP32_MFMA_PEAK_TF,
            "unit": "TFLOP/s", "frac": round(tf / FP32_MFMA_PEAK_TF, 4), "avg_launch_us": round(ms * 1e3, 2)}


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
    ap.add_argument("--full", action="store_true",
                    help="configs[2] per-GPU work: add the 3D->2D dense head loss (default: configs[1])")
    a = ap.parse_args()

    from moleculesde_amd import _lib, dp, pretrain
    from moleculesde_amd.geom3d import prepare_batch
    from moleculesde_amd.synthetic import make_batch, batch_stats
    _lib.load()
    rank, world, local = dp.init_from_env("cuda")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if world != a.gpus and rank == 0:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    torch.manual_seed(0)

    args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=1 if a.full else 0, batch_size=a.batch_size)
    trainer = pretrain.Trainer(args, device)
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
        for b in pool:
            trainer.capture(b)
        for b in pool:
            trainer.step_graph(b)
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
        roof = roofline_cfconv(trainer, pool[0])
        fused = roofline_fused(trainer, pool[0])
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
            "roofline_fused_cfconv_fwd": fused,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.batch_size)
        print(json.dumps(out), flush=True)
    dp.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
