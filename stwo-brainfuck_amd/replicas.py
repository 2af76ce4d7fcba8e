"""Multi-GPU host logic (DESIGN.md section multi-GPU): one process per GPU. Replicas (default): each rank proves its own independent
trace, no data-path collective (the reference has no multi-device path at all — SURVEY.md section 2.3). Shard group: the ranks prove
ONE trace together; the data-path exchanges are issued by libbfhip itself (RCCL on the context's stream) and torch.distributed only
carries the timing protocol (barrier on both sides of the timed region, MAX over ranks of the elapsed time) and the 128-byte unique id."""
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


def share_unique_id(dist, make_id, device=None) -> bytes:
    """Control plane of a multi-process shard group (Context.join_rccl_group): rank 0 creates the 128-byte RCCL unique id with make_id()
    and every rank receives it through one broadcast of the already initialised torch.distributed group (RCCL tensors when `device` is
    a CUDA device, gloo on CPU tensors otherwise). The proof's data never travels this way: the library issues its own collectives."""
    import torch
    rank = dist.get_rank()
    t = torch.zeros(128, dtype=torch.uint8)
    failure = None
    if rank == 0:
        # a rank 0 that cannot create the id (librccl missing ...) must still take part in the broadcast: the other ranks are already waiting in it.
        # It sends 128 zero bytes — never a valid id — and every rank raises.
        try:
            raw = make_id()
            if len(raw) != 128:
                raise ValueError("an RCCL unique id has 128 bytes")
            t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).clone()
        except Exception as e:
            failure = e
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=0)
    if failure is not None:
        raise failure
    out = bytes(t.cpu().numpy().tobytes())
    if out == bytes(128):
        raise RuntimeError("rank 0 could not create the RCCL unique id (it broadcast the all-zero marker)")
    return out
