"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" on CPU for tests).

The hot path shards over PICTURES: pictures that do not reference each other -- all pictures of an
intra-only stream, separate closed GOPs, the pictures of one temporal level of a hierarchical GOP -- are reconstructed by
different ranks with no collective in the data path.  The one exchange the path can need is the delivery of a finished
reference picture to the ranks that predict from it: a broadcast from its owner (frame_parallel.py)."""
import os
import time


def init_from_env(backend=None):
    """returns (dist module or None, rank, world, local_rank); rendezvous over 127.0.0.1 by default"""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and not os.environ.get("HMGPU_FORCE_DIST"):
        return None, 0, 1, local_rank
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("NCCL_DEBUG", "WARN")          # keep RCCL's version banner off stdout (bench.py prints ONE JSON line)
    if backend is None:
        backend = os.environ.get("HMGPU_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if os.environ.get("HMGPU_SINGLE_DEVICE"):        # rehearsal of N ranks on a one-GPU box (gloo only: RCCL refuses two ranks on one GPU)
        local_rank = 0
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kw["device_id"] = torch.device("cuda", local_rank)
    # RCCL prints a version banner on stdout when the communicator comes up; bench.py owns stdout (ONE JSON line), so the
    # banner is sent to stderr: fd 1 is pointed at fd 2 while the group and its first collective are created
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        dist.barrier()
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    return dist, rank, world, local_rank


def shard(num_units, world, rank):
    """contiguous share of `num_units` independent units (pictures / GOPs) for `rank`: [begin, end)"""
    base, rem = divmod(num_units, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def reduce_device(dist, device):
    """where the MAX-over-ranks scalar lives: on the GPU for RCCL, on the host for gloo"""
    return device if (dist is not None and dist.get_backend() == "nccl") else "cpu"


def timed_region(dist, run, sync, device=None):
    """barrier + sync, run(), sync + barrier; returns the MAX elapsed seconds over all ranks (the contract of bench.py)"""
    sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    run()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device(dist, device) if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    return elapsed
