#!/usr/bin/env python3
"""Benchmark of the hot path: prove_brainfuck on MI355X, metric = trace cells committed+proved per second (BASELINE.json).

A "step" is one complete proof (preprocessed commitment .. decommitment, crates/brainfuck_prover/src/brainfuck_air/mod.rs:493-734)
of one trace whose row-granular table columns are already resident in HBM. Workload at N=1: BASELINE.json configs[1]
(fib19.bf, largest component 2^20 table rows = 2^24 domain rows, Blake2s Merkle, LOG_MAX_ROWS = 24).
N > 1: one process per GPU. Default: every rank proves its own independent trace (replicas, weak scaling, no data-path collective).
--shard: the N ranks prove ONE trace together (shard group, strong scaling).

Prints ONE JSON line on rank 0 (driver contract). The roofline object is measured live with HIP events on the library's stream;
the cpu_baseline object times the CPU oracle ("port") on a bounded sample on the host cores of the same box.
"""
import argparse
import ctypes
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

BLAKE2S_PEAK_GCPS = 39.9   # tools/ubench_blake.hip on MI355X: register-only compression loop, all CUs
# Blake2s compressions per fib19 proof at LOG_MAX_ROWS = 24 after replication-aware dedup (counted by tools/count_compressions.py)
MERKLE_COMPRESSIONS_FIB19_LMR24 = 674228124

FIB19 = "+++++++++++++++++>+>+<<[->>[->+>+<<]<[->>+<<]>>[-<+>]>[-<<<+>>>]<<<<]>>."  # tests/golden/programs/fib19.bf (workload input)


def load_package():
    name = "stwo_brainfuck_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "stwo-brainfuck_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def cpu_baseline():
    """Times the CPU oracle (kind "port": the Rust reference cannot be built on this image) on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import Oracle
    orc = Oracle()
    path = os.path.join(ROOT, "tests", "golden", "programs", "collatz.bf")
    code = open(path).read()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    log_sizes, steps = orc.log_sizes(code, b"7\n")
    main_cols = [8, 8, 4, 9, 13, 13, 11, 11, 11, 11, 11, 11, 7]
    inter_cols = [4, 4, 4, 12, 4, 4, 4, 4, 4, 4, 4, 4, 4]
    cells = sum((m + i) << l for m, i, l in zip(main_cols, inter_cols, log_sizes))
    # the port's OpenMP loops stop scaling at a few dozen threads: try a few team sizes and report the best one
    best_sec, best_threads, tried = None, 1, []
    for threads in sorted({min(avail, t) for t in (16, 32, 64)}):
        orc.L.orc_set_threads(threads)
        _, _, sec = orc.prove(code, b"7\n", log_max_rows=max(log_sizes))
        tried.append(f"{threads}t {sec:.1f}s")
        if best_sec is None or sec < best_sec:
            best_sec, best_threads = sec, threads
    note = ""
    try:   # the same port on the benchmark workload itself, measured once when the parity digest was generated
        fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fib19_lmr24_oracle_proof.json")))
        note = f"; the same port needed {fx['oracle_seconds']:.0f} s for the fib19 workload itself on an 8-core host (tests/golden/fib19_lmr24_oracle_proof.json)"
    except Exception:
        pass
    return {"value": cells / best_sec, "unit": "trace cells/s", "cores": best_threads, "kind": "port",
            "sample": f"collatz.bf input '7\\n' ({steps} VM steps, {cells} cells, LOG_MAX_ROWS={max(log_sizes)}), one proof per OpenMP team size "
                      f"({', '.join(tried)}), best reported" + note}


def pick_device(local_rank, n_visible, override=None):
    """One process per GPU: rank r drives device LOCAL_RANK. A launcher that narrows each rank's view to its own GPU
    (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank) leaves one visible device, numbered 0, on every rank."""
    if override is not None:
        return override
    return local_rank if local_rank < n_visible else local_rank % max(n_visible, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-max-rows", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--kernel-events", default="dominant", choices=["dominant", "all"], help="HIP-event timing of the dominant kernel only (default, ~1%% overhead) or of every kernel (~10%%)")
    ap.add_argument("--reuse-preprocessed", action="store_true", help="NOT the headline: keep the program-independent preprocessed tree across proofs (a deployment option; the reference recommits it per proof)")
    ap.add_argument("--inflight", type=int, default=1, help="proofs in flight per GPU (one host thread + HIP stream each); >1 reports pipelined throughput, no roofline")
    ap.add_argument("--shard", action="store_true", help="NOT the default: with --gpus N > 1 the N ranks prove ONE trace together (shard group: share-wise Merkle layers, "
                    "one all-gather per tree, one max-reduce per proof; strong scaling of a single proof) instead of N independent replicas")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) | gloo (test mode on boxes with fewer GPUs than ranks)")
    ap.add_argument("--device", type=int, default=None, help="test mode: every rank uses this device instead of LOCAL_RANK")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        device = pick_device(local_rank, torch.cuda.device_count(), args.device)
        torch.cuda.set_device(device)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    pkg = load_package()
    if pkg.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    device = pick_device(local_rank, pkg.device_count(), args.device)
    ctx = pkg.Context(device, max_log_domain=args.log_max_rows + 2)   # one process per GPU: rank r drives device LOCAL_RANK
    trace = pkg.Trace(ctx, FIB19, b"")          # VM + table build + upload: outside the timed region (inputs resident in HBM)
    lib = pkg.lib()
    if args.reuse_preprocessed:
        lib.bfhip_ctx_reuse_preprocessed(ctx._h, 1)
    extra = []
    if args.inflight > 1:
        args.no_kernel_events = True            # the event profiler is per process, not per stream
        for _ in range(args.inflight - 1):
            c2 = pkg.Context(device, max_log_domain=args.log_max_rows + 2)
            extra.append((c2, pkg.Trace(c2, FIB19, b"")))

    def one_step():
        if not extra:
            return trace.prove(args.log_max_rows)
        import threading
        res = [None] * (1 + len(extra))
        def run(i, tr):
            res[i] = tr.prove(args.log_max_rows)
        th = [threading.Thread(target=run, args=(i + 1, tr)) for i, (_, tr) in enumerate(extra)]
        for t in th:
            t.start()
        run(0, trace)
        for t in th:
            t.join()
        return res[0]

    spec = importlib.util.spec_from_file_location("stwo_brainfuck_amd_replicas", os.path.join(ROOT, "stwo-brainfuck_amd", "replicas.py"))
    replicas = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(replicas)
    sharded = args.shard and dist is not None and world > 1
    if sharded:
        if args.inflight > 1:
            raise SystemExit("--shard and --inflight are exclusive")
        dev = torch.device("cuda", device) if args.dist_backend == "nccl" else None
        ctx.set_shard(rank, world, *replicas.shard_exchanges(dist, dev))

    def sync():
        ctx.sync()
        for c2, _ in extra:
            c2.sync()
        torch.cuda.synchronize()

    def start_events():
        if not args.no_kernel_events:
            lib.bfhip_profile_enable(ctx._h, 1 if args.kernel_events == "all" else 2)
            lib.bfhip_profile_reset(ctx._h)

    cuda_t = (lambda v: torch.tensor([v], dtype=torch.float64, device="cuda")) if (dist is not None and args.dist_backend == "nccl") else None
    dt, (proof, phases) = replicas.timed_region(one_step, args.steps, args.warmup, dist=dist, sync_fn=sync,
                                                backend_tensor=cuda_t, on_timed_start=start_events)
    # replicas: every rank proves its own trace (units add up); shard group: all ranks prove the same one
    total_cells = trace.cells if sharded else replicas.aggregate_units(trace.cells * args.inflight, dist=dist, backend_tensor=cuda_t)

    roofline = None
    if not args.no_kernel_events:
        js = ctypes.c_void_p()
        lib.bfhip_profile_report(ctx._h, ctypes.byref(js))
        rep = json.loads(ctypes.string_at(js).decode())
        lib.bfhip_free_host(js)
        lib.bfhip_profile_enable(ctx._h, 0)
        name, d = max(rep.items(), key=lambda kv: kv[1]["total_ms"])
        avg_ms = d["total_ms"] / d["calls"]
        achieved = d["bytes"] / d["calls"] / (avg_ms * 1e-3) / 1e9
        traffic = None
        import glob
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))   # from the separate rocprofv3 --pmc passes of the latest round
        if pmc_files:
            traffic = json.load(open(pmc_files[-1])).get(name, {}).get("hbm_bytes_per_launch")
        roofline = {"kernel": name, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "launches": d["calls"], "avg_launch_us": round(avg_ms * 1e3, 2),
                    "algorithmic_bytes_per_launch": round(d["bytes"] / d["calls"]),
                    "kernels_ms_per_step": {k: round(v["total_ms"] / args.steps, 3) for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"])},
                    "kernels_GBps_moved": {k: round(v["bytes"] / v["total_ms"] / 1e6, 1) for k, v in rep.items() if v["bytes"] > 0}}
        if name == "k_merkle_layer":
            # The Merkle kernel is integer-VALU bound, not HBM bound: one Blake2s compression (~977 VALU ops) per 64 message bytes.
            # Peak = 39.9 G compressions/s measured with tools/ubench_blake.hip (registers only) on the same chip.
            comp = MERKLE_COMPRESSIONS_FIB19_LMR24 if args.log_max_rows == 24 else None
            if comp:
                rate = comp * args.steps / (d["total_ms"] * 1e-3) / 1e9
                roofline["valu"] = {"unit": "G Blake2s compressions/s", "achieved": round(rate, 2), "peak_measured": BLAKE2S_PEAK_GCPS, "frac": round(rate / BLAKE2S_PEAK_GCPS, 3),
                                    "compressions_per_proof": comp}

    cells = trace.cells
    if rank == 0:
        out = {
            "metric": "trace cells committed+proved/sec",
            "value": total_cells * args.steps / dt,
            "unit": "trace cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "u32 (M31 / QM31 modular arithmetic)",
            "data": "fib19.bf execution trace (199246 VM steps); proof bytes identical to the CPU oracle's proof of this workload (tests/golden/fib19_lmr24_oracle_proof.json)",
            "config": {"workload": "fib19.bf, largest component 2^20 table rows = 2^24 domain rows, Blake2s Merkle, 1 proof per step",
                       "log_max_rows": args.log_max_rows, "cells_per_proof": cells, "main_cells": trace.main_cells, "interaction_cells": trace.interaction_cells,
                       "component_log_sizes": trace.log_sizes, "parallelism": ("shard group: one proof over all ranks, share-wise Merkle layers" if sharded else "replicas") if world > 1 else "single", "proofs_in_flight_per_gpu": args.inflight, "preprocessed_tree": "reused across proofs" if args.reuse_preprocessed else "recommitted every proof (as the reference)",
                       "proof_bytes": len(proof), "phase_ms_last_step": {k: round(v * 1e3, 2) for k, v in phases.items()}},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if args.reuse_preprocessed:
        lib.bfhip_ctx_reuse_preprocessed(ctx._h, 0)
    if sharded:
        ctx.set_shard(0, 1)
    for c2, t2 in extra:
        t2.close(); c2.close()
    trace.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
