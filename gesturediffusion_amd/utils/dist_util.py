"""Device selection + one-process-per-GPU helpers.

`setup_dist(device)` / `dev()` keep the reference's single-device semantics
(`utils/dist_util.py:18-51`).  The reference never initialises a process group (its MPI/NCCL code
is commented out, `:26-41`); multi-GPU here is new functionality around the hot path: samples of
a batch are independent, so each rank samples a contiguous shard with zero communication and one
RCCL gather over xGMI at the end returns the batch to rank 0.  N > 1 has only been
exercised on CPU (gloo, world size 2) and as several ranks sharing one GPU; RCCL across GPUs is the driver's run.
`check_world(total)` is the callers' up-front refusal of more ranks than samples.
"""
import os

import torch as th
import torch.distributed as dist

GPUS_PER_NODE = 8
used_device = 0


def setup_dist(device=0):
    global used_device
    used_device = device


def dev():
    if th.cuda.is_available() and used_device >= 0:
        return th.device(f"cuda:{used_device}")
    return th.device("cpu")


def load_state_dict(path, **kwargs):
    kwargs.setdefault("weights_only", True)
    return th.load(path, **kwargs)


# --------------------------------------------------------------------------- multi-GPU sharding
def init_from_env(backend=None):
    """One process per GPU (torchrun / torch.distributed.run env).  Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = th.cuda.is_available()
    if use_cuda and os.environ.get("GDX_SINGLE_GPU_RANKS"):   # rehearsal: several ranks share GPU 0 (gloo backend)
        local = 0
    device = th.device(f"cuda:{local}") if use_cuda else th.device("cpu")
    if use_cuda:
        th.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("GDX_DIST_BACKEND") or ("nccl" if use_cuda else "gloo")
        if backend == "nccl":
            # device_id binds the rank to its GPU and creates the RCCL communicator now, with every rank taking part:
            # the path's only transfer is a point-to-point group in which ranks may sit out (gather_samples)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    setup_dist(local if use_cuda else -1)
    return rank, world, device


def check_world(total, world=None):
    """Every rank must own at least one sample (an idle rank would sit out of the RCCL gather and fail gdx_prepare(0)):
    raised identically on every rank, before any of them has touched the data path."""
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    if total < world:
        raise ValueError(f"{total} samples cannot be sharded over {world} ranks: every rank needs at least one "
                         f"(use --gpus / --nproc-per-node <= {total})")


def shard_range(total, rank, world):
    """Contiguous shard [lo, hi) of `total` samples for `rank`; sizes differ by at most one."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_samples(local, total, dst=0):
    """End-of-loop gather of per-rank sample shards [b_r, ...] to `dst` -- the path's only collective.

    One group of point-to-point transfers: `dst` posts a receive per peer straight into that peer's row range of ONE
    preallocated [total, ...] buffer (row ranges of a contiguous tensor are contiguous, so nothing is staged, padded or
    concatenated), every other rank posts one send of its shard.  On GPUs this is an ncclSend / ncclRecv group over the
    direct xGMI links: `total` samples cross the fabric once and only `dst` holds the full batch (an all_gather moves
    world x that and materialises it on every rank).  Uneven shards (41 samples over 8 ranks) need no padding.  A rank
    with an empty shard (total < world) posts nothing: fine on gloo, but RCCL wants every rank of a group in its first
    transfer, so on the nccl backend that case is an error on ALL ranks (the callers refuse it up front, before any rank
    has started sampling).  Returns the full batch on `dst`, None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    if local.is_cuda and dist.get_backend() != "nccl":
        # rehearsal backends (gloo) move host memory and know nothing about HIP streams: stage through the host
        # (.cpu() waits for the sampling loop that is still in flight on the current stream)
        full = gather_samples(local.cpu(), total, dst)
        return full.to(local.device) if full is not None else None
    if total < world and dist.get_backend() == "nccl":
        raise ValueError(f"{total} samples over {world} ranks leaves ranks without a sample: use at most {total} ranks")
    spans = [shard_range(total, r, world) for r in range(world)]
    lo, hi = spans[rank]
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} samples, its shard of {total} is {hi - lo}")
    full, ops = None, []
    if rank == dst:
        full = local.new_empty((total,) + tuple(local.shape[1:]))
        full[lo:hi].copy_(local)
        ops = [dist.P2POp(dist.irecv, full[a:b], r) for r, (a, b) in enumerate(spans) if r != dst and b > a]
    elif hi > lo:
        ops = [dist.P2POp(dist.isend, local.contiguous(), dst)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return full
