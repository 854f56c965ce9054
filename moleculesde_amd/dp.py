"""Data parallelism: one process per GPU, `torch.distributed` (backend "nccl" == RCCL on ROCm, over
xGMI), molecules sharded across ranks, ONE all-reduce of the flat fp32 gradient buffer per step
(4.67 M elements = 18.7 MB).  The reference has no distributed code; semantics chosen
here: BatchNorm statistics are per replica, contrastive negatives are permuted within a replica,
losses are averaged over ranks for logging.  Works on CPU tensors with the gloo backend, which is
how the N>1 path is tested without GPUs."""
import os

import torch
import torch.distributed as dist

from . import wcache


FORCE_COLLECTIVES = False    # debug: run the collectives even at world size 1 (exercises RCCL on one GPU)


def init_from_env(device_type="cuda", force=False):
    """Initialise the process group from torchrun's environment; returns (rank, world, local_rank)."""
    global FORCE_COLLECTIVES
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if force:
        FORCE_COLLECTIVES = True
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # MSDE_DP_BACKEND=gloo on device tensors: how the N>1 trainer step is tested with several ranks sharing ONE
        # GPU (RCCL refuses two ranks on one device)
        backend = os.environ.get("MSDE_DP_BACKEND") or ("nccl" if device_type == "cuda" else "gloo")
        if device_type == "cuda" and backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def broadcast_flat(flat, src=0):
    """Make every replica start from rank `src`'s parameters."""
    if world_size() > 1 or (FORCE_COLLECTIVES and dist.is_initialized()):
        dist.broadcast(flat, src=src)
        if flat.is_cuda:                  # parameters rewritten behind autograd's and the optimiser's back
            wcache.invalidate_weight_copies()
    return flat


def allreduce_mean_(flat):
    """Sum the flat gradient buffer over ranks (in place); returns the scale (1/world) that the
    optimiser kernel applies, so no extra pass over the buffer is spent on the division."""
    w = world_size()
    if w > 1 or (FORCE_COLLECTIVES and dist.is_initialized()):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return 1.0 / w


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_buckets_async(flat, ranges, order=None):
    """One asynchronous all-reduce (sum) per bucket of the flat gradient buffer, issued in `order` (default: the
    reverse of the buffer order = the order in which the backward pass completes the models' gradients: heads
    first, encoders last).  Returns (order, works, scale): wait on works[k] before touching bucket order[k]; the
    waits are stream waits (RCCL) -- the host does not block -- so the optimiser kernel of one bucket runs while
    the next bucket is on the wire.  xGMI ring collectives are per-link bound: 2-4 buckets of >= 1 MB keep each
    message in the bandwidth regime while hiding all but the first bucket's latency."""
    w = world_size()
    order = list(range(len(ranges) - 1, -1, -1)) if order is None else list(order)
    works = []
    live = w > 1 or (FORCE_COLLECTIVES and dist.is_initialized())
    for i in order:
        a, b = ranges[i]
        works.append(dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, async_op=True) if (live and b > a) else None)
    return order, works, 1.0 / w


def shard_seed(base_seed, rank):
    """Rank-distinct seed for synthetic shards / shuffling."""
    return int(base_seed) * 1000003 + int(rank)


def barrier():
    if world_size() > 1:
        dist.barrier()
