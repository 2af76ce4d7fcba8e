"""Multi-GPU host logic for the "replicas only" mode: one process per GPU, each rank proves its own
independent trace; there is no data-path collective (the reference has no multi-device path at all — SURVEY.md §2.3).
Only the timing protocol needs torch.distributed: barrier on both sides of the timed region and MAX over ranks of the elapsed time."""
import time


def rank_device(local_rank: int, device_count: int) -> int:
    """One process per GPU: rank r of a node drives device r (LOCAL_RANK)."""
    if device_count < 1:
        raise RuntimeError("no GPU visible: the HIP backend has no CPU fallback")
    if local_rank >= device_count:
        raise RuntimeError(f"LOCAL_RANK {local_rank} but only {device_count} devices")
    return local_rank


def timed_region(step_fn, steps: int, warmup: int, dist=None, sync_fn=None, backend_tensor=None, on_timed_start=None):
    """Runs `warmup` untimed then exactly `steps` timed calls of step_fn(); returns (max-over-ranks seconds, last step result).
    dist: an initialised torch.distributed module or None (single process). sync_fn: device synchronisation (both sides)."""
    def fence():
        if dist is not None:
            dist.barrier()
        if sync_fn is not None:
            sync_fn()

    last = None
    for _ in range(warmup):
        last = step_fn()
    if on_timed_start is not None:
        on_timed_start()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step_fn()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64) if backend_tensor is None else backend_tensor(dt)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, last


def aggregate_units(units_this_rank: int, dist=None, backend_tensor=None) -> int:
    """Total units (trace cells) processed per step by all ranks (weak scaling: every rank proves its own trace)."""
    if dist is None:
        return units_this_rank
    import torch
    t = torch.tensor([float(units_this_rank)], dtype=torch.float64) if backend_tensor is None else backend_tensor(float(units_this_rank))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def shard_exchanges(dist, device=None):
    """The two exchanges of a shard group (Context.set_shard) over torch.distributed: RCCL when `device` is a CUDA device
    (backend "nccl"), gloo on CPU tensors otherwise. Returns (allgather, allreduce_max)."""
    import numpy as np
    import torch

    world = dist.get_world_size()

    def allgather(send: bytes) -> bytes:
        t = torch.frombuffer(bytearray(send), dtype=torch.uint8)
        if device is not None:
            t = t.to(device)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return b"".join(o.cpu().numpy().tobytes() for o in outs)

    def allreduce_max(values):
        t = torch.from_numpy(values.astype(np.int64))     # u32 values as int64: MAX is then the unsigned maximum on every backend
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.cpu().numpy().astype(np.uint32)

    return allgather, allreduce_max
