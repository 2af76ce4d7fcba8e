#!/usr/bin/env python3
"""Benchmark of the hot path: prove_brainfuck on MI355X, metric = trace cells committed+proved per second (BASELINE.json).

A "step" is one complete proof (preprocessed commitment .. decommitment, crates/brainfuck_prover/src/brainfuck_air/mod.rs:493-734)
of one trace whose row-granular table columns are already resident in HBM. Workload at N=1: BASELINE.json configs[1]
(fib19.bf, largest component 2^20 table rows = 2^24 domain rows, Blake2s Merkle, LOG_MAX_ROWS = 24).
N > 1: one process per GPU, and the headline is STRONG scaling: the N ranks prove ONE trace of BASELINE config 4's size together (fib19.bf:
2^24 domain rows, Blake2s — the N = 1 workload, so the curve is like for like) as a shard group over RCCL; the
N independent replicas (weak scaling, no data-path collective) ride in the `replicas` field. --replicas makes them the headline instead.
`python3 bench.py --gpus N` starts the N ranks ITSELF when no launcher did (WORLD_SIZE unset): fresh child processes, before this process
touches the GPU; under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks are the launcher's. --gpus and
WORLD_SIZE must agree.

This file holds the contract path only (W warm-up proofs, barrier, K timed proofs, barrier, MAX over ranks, ONE JSON line on rank 0); everything
that rides beside it lives in tools/benchlib/ (one paragraph per module in its __init__.py). Checked and measured inside this run:
  parity_checked  SHA-256 of the last timed proof == the committed digest of the CPU oracle's proof of the same workload
  roofline        dominant kernel, HIP events on the library's own stream over the timed region (k_merkle_layer: integer-VALU bound, Blake2s compressions
                  counted from the launch shapes; HBM figure beside it); sustained_clock_ghz / frac_at_sustained_clock from the library's in-run clock probe
  config          (the driver's record keeps `config` whole) metric_point = the metric's own size, 2^22 rows; batch = that size through the library's pool,
                  one caller thread; sweep_ms = ms per proof at 2^20..2^26 rows
  fft             the circle-FFT kernels against both of their bounds (HBM bytes moved; VALU lane-ops), from one extra untimed, fully instrumented proof
  sweep, poseidon252, pipelined   the details behind config.*: synthetic traces 2^20..2^26, BASELINE config 5 on one GPU, proofs in flight per size
  strong_scaling  N > 1 only: per workload ms_per_proof over the group, the one-GPU time of the same proof in the same run, speedup_vs_n1, comm share
  cpu_baseline    the CPU oracle ("port") — tools/benchlib/cpu.py
"""
import argparse
import hashlib
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from tools.benchlib import group as group_mod                                                                                    # noqa: E402
from tools.benchlib.cpu import cpu_baseline, cpu_baseline_block, host_cpu_budget, simd_bound, simdbackend_work_counts             # noqa: E402,F401
from tools.benchlib.launcher import launch_ranks, pinned_host_thread, rank_environments                                         # noqa: E402,F401
from tools.benchlib.probes import batch_summary, pipelined_child_main, run_pipelined, run_poseidon_point, run_sweep              # noqa: E402
from tools.benchlib.roofline import add_sustained_clock, dominant_roofline, fft_report, profile_report, roofline_from_committed_rocprof   # noqa: E402,F401
from tools.benchlib.shard import run_shard_probe, shard_probe                                                                    # noqa: E402
from tools.benchlib.workloads import (FIB19, HBM_PEAK_GBS, VALU_OPS_PER_COMPRESSION, VALU_PEAK_TOPS, committed_digests, kernel_sources_sha256,   # noqa: E402,F401
                                      load_package, pick_device, sweep_program, want_digest)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one process per GPU). Without a launcher (WORLD_SIZE unset) and N > 1 this process starts the N ranks itself; "
                    "under a launcher it must equal WORLD_SIZE. Default: WORLD_SIZE, or 1")
    ap.add_argument("--launch-timeout", type=int, default=1500, help="self-launcher: seconds the N ranks may take before they are ended")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-max-rows", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pin", action="store_true", help="pin the proving host thread to one core of the GPU's NUMA node during the timed regions (default off: the gain "
                    "is box dependent, profiles/r03_bench_host_pinning_ab.txt)")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "full", "sample"],
                    help="full: time the CPU port on the bench workload itself live on this box (~20 s, ~20 GB of host memory); sample: committed "
                         "full-size measurement + a bounded live sample; auto (default): full when the host has >= 32 cores and >= 48 GB of free memory")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--kernel-events", default="dominant", choices=["dominant", "all"], help="HIP-event timing of the dominant kernel only (default, ~0.5%% overhead) or of every kernel (~10%%) inside the timed region")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the 0.6 s sustained-clock probe behind the timed region (roofline.sustained_clock_ghz)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the 2^20..2^26 synthetic sweep and the proofs-in-flight children (N=1 only; ~30 s)")
    ap.add_argument("--sweep-steps", type=int, default=3)
    ap.add_argument("--no-poseidon", action="store_true", help="skip the Poseidon252 2^26-row point (BASELINE config 5 on one GPU; ~15 s)")
    ap.add_argument("--poseidon-log", type=int, default=26)
    ap.add_argument("--sweep-logs", default="20,21,22,23,24,25,26")
    ap.add_argument("--conventions", default="0,0,0,0", help="merkle_node_hash,mix_u64,logup_mask_order,merkle_channel (include/bfhip.h bfhip_conventions); default = stwo defaults, Blake2s channel")
    ap.add_argument("--reuse-preprocessed", action="store_true", help="NOT the headline: keep the program-independent preprocessed tree across proofs (a deployment option; the reference recommits it per proof)")
    ap.add_argument("--inflight", type=int, default=1, help="proofs in flight per GPU (one host thread + context each); >1 reports pipelined throughput, no roofline")
    ap.add_argument("--shard", action="store_true", help="(default for N > 1 since round 5; kept for old command lines) the N ranks prove ONE trace together")
    ap.add_argument("--replicas", action="store_true", help="N > 1: headline = N independent proofs (weak scaling) instead of ONE proof over the shard group (strong scaling, default)")
    ap.add_argument("--no-extra-stages", action="store_true", help="N > 1: only the headline workload over the group, not the 2^24-row synthetic trace (configs 3/4) and the 2^26-row Poseidon252 trace (config 5)")
    ap.add_argument("--shard-policy", type=int, default=-1, choices=[-1, 0, 1], help="N > 1: bfhip_ctx_set_shard_policy of every rank's contexts: 0 = exchange columns -> rows (column-sharded transforms), 1 = replicate "
                    "the transforms (only hashing / quotients / folds divided), -1 = automatic (default: replicate for a group of two ranks on different GPUs)")
    ap.add_argument("--group-inflight", type=int, default=1, help="N > 1: after the headline, K proofs in flight over the shard group (K contexts and host threads per rank, K communicators): "
                    "strong_scaling.workloads.fib19_K_in_flight. Default 1 = off (two communicators driven concurrently are unmeasured with librccl itself)")
    ap.add_argument("--group-timeout", type=int, default=900, help="N > 1: seconds the shard group's headline part (join, timed proofs) and, separately, its extra stages may take; after that rank 0 prints "
                    "what was measured before (the replicas line / the strong-scaling line without the extra stages) and every rank leaves with exit code 3 (a collective that never returns cannot be interrupted)")
    ap.add_argument("--rccl-child-probe", action="store_true", help="N > 1: also run the group stages in child processes (one per rank, RCCL) before the ranks touch their GPUs (debugging a transport that takes the main process down)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) | gloo (test mode on boxes with fewer GPUs than ranks)")
    ap.add_argument("--device", type=int, default=None, help="test mode: every rank uses this device instead of LOCAL_RANK")
    ap.add_argument("--no-shard-probe", action="store_true", help="N > 1: skip the child-process probes altogether (same as --no-local-probe without --rccl-child-probe)")
    ap.add_argument("--probe-steps", type=int, default=8)
    ap.add_argument("--probe-timeout", type=int, default=240)
    ap.add_argument("--probe-fib19-only", action="store_true", help="shard probe: only the bench workload, not the 2^24-row and 2^26-row Poseidon252 traces (BASELINE configs 3-5)")
    ap.add_argument("--no-local-probe", action="store_true", help="shard probe: skip the in-process variant (rank 0's child driving all N GPUs from N host threads)")
    for hidden in ("--pipelined-child", "--probe-out", "--probe-id-file"):
        ap.add_argument(hidden, default=None, help=argparse.SUPPRESS)
    for hidden in ("--probe-local", "--shard-probe"):
        ap.add_argument(hidden, action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def main():
    args = parse_args()
    if args.shard_probe:
        return shard_probe(args)
    if args.pipelined_child:
        return pipelined_child_main(args)
    if "WORLD_SIZE" not in os.environ and (args.gpus or 1) > 1:
        # no launcher started the ranks: do it here, before anything in this process touches the GPU (children are fresh processes)
        return launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus, args.launch_timeout)
    rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is not None and args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: the two must agree "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}, or plain python3 bench.py --gpus {args.gpus})")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1:
        # a rank that is ended from outside (the self-launcher's time limit, a driver's timeout) says where it was: Python stacks of all its threads
        import faulthandler
        import signal
        faulthandler.register(signal.SIGTERM, all_threads=True, chain=True)
        os.environ.setdefault("BFHIP_COMM_TIMEOUT_S", "120")     # a collective that never completes ends the stage with an error, not the run with a hang

    # ---- child processes that must start BEFORE this process touches the GPU ---------------------------------------------------------------------------
    # N > 1: ONE process driving all N GPUs over the in-process transport (a second transport on the same hardware); bounded, killed on timeout.
    shard_probe_result = None
    if world > 1 and not args.replicas and not args.no_shard_probe and (args.rccl_child_probe or not args.no_local_probe) and world & (world - 1) == 0:
        try:
            shard_probe_result = run_shard_probe(args, rank, world)
        except Exception as e:
            shard_probe_result = {"n_gpus": world, "error": repr(e)}
    # N = 1: proofs in flight through the library's pool, {fib19, 2^22 rows (the metric's size), 2^20 rows} x {1, 2, 3 in flight}, one configuration per
    # child. Reported beside `value`, never as it: value is one proof at a time (one prove_brainfuck call = one proof, mod.rs:471-735).
    pipelined = None
    if world == 1 and args.inflight == 1 and not args.no_sweep:
        try:
            pipelined = run_pipelined(args)
        except Exception as e:
            pipelined = {"error": repr(e)}

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
    R = group_mod.Ranks(args, rank, local_rank, world, torch, dist)

    pkg = load_package()
    if pkg.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    conv = tuple((([int(v) for v in args.conventions.split(",")]) + [0, 0, 0, 0])[:4])
    pkg.set_default_conventions(*conv)
    device = pick_device(local_rank, pkg.device_count(), args.device)
    ctx = pkg.Context(device, max_log_domain=args.log_max_rows + 2)   # one process per GPU: rank r drives device LOCAL_RANK
    trace = pkg.Trace(ctx, FIB19, b"")          # VM + table build + upload: outside the timed region (inputs resident in HBM)
    lib = pkg.lib()
    if args.reuse_preprocessed:
        lib.bfhip_ctx_reuse_preprocessed(ctx._h, 1)
    extra = []
    if args.inflight > 1:
        args.no_kernel_events = True            # pipelined throughput run: no per-kernel figures
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
        [t.start() for t in th]
        run(0, trace)
        [t.join() for t in th]
        return res[0]

    def sync():
        ctx.sync()
        for c2, _ in extra:
            c2.sync()
        torch.cuda.synchronize()

    def start_events():
        if not args.no_kernel_events:
            lib.bfhip_profile_enable(ctx._h, 1 if args.kernel_events == "all" else 2)
            lib.bfhip_profile_reset(ctx._h)

    spec = importlib.util.spec_from_file_location("stwo_brainfuck_amd_replicas", os.path.join(ROOT, "stwo-brainfuck_amd", "replicas.py"))
    replicas = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(replicas)
    pin = lambda: pinned_host_thread(args.pin and args.inflight == 1, device, local_rank, world)      # noqa: E731

    # ---- the line: built from `S` (everything measured so far), so that the extra-stages watchdog of an N > 1 run can still print what was measured ---------
    S = {"sweep": None, "poseidon": None, "fft": None, "roofline": None, "cpu_baseline": None}

    def build_line(S, g, extras_error=None):
        sharded, dt, proof, phases = S["sharded"], S["dt"], S["proof"], S["phases"]
        digest = hashlib.sha256(proof).hexdigest()
        want = want_digest(conv, args.log_max_rows)
        parity_checked = bool(want is not None and want["sha256"] == digest and want["proof_bytes"] == len(proof))
        verified, _ = pkg.verify_brainfuck(proof, args.log_max_rows)
        sweep = S["sweep"]
        p22 = next((p for p in sweep if p["log_domain_rows"] == 22), None) if isinstance(sweep, list) else None
        p22_name = "synthetic nested-counter trace, Memory component 2^22 domain rows, LOG_MAX_ROWS 22 (BASELINE metric 'at 2^22 rows')"
        ms_step = dt / args.steps * 1e3
        metric_point = ({"workload": p22_name, "rows": "2^22", "value": p22["cells_per_s"], "cells_per_s": p22["cells_per_s"], "unit": "trace cells/s", "ms_per_proof": p22["ms_per_proof"],
                         "cells": p22["cells"], "proof_sha256": p22["proof_sha256"], "sha256": p22["proof_sha256"], "verified": p22["verified"],
                         "pipelined": ({k: {kk: vv for kk, vv in v.items() if kk not in ("proof_sha256", "how")} for k, v in pipelined["2^22_rows"].items() if k.startswith("in_flight_")}
                                       if isinstance(pipelined, dict) and "2^22_rows" in pipelined else None)} if p22 else None)
        out = {
            "metric": "trace cells committed+proved/sec", "value": S["total_cells"] * args.steps / dt, "unit": "trace cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "metric_point": metric_point,      # BASELINE.json's metric is quoted "at 2^22 rows" (value above is the larger fib19 workload); also inside `config`, which the driver keeps whole
            "higher_is_better": True, "scaling": "strong" if sharded else "weak",
            "speedup_vs_n1": (round(g["n1"]["ms_per_proof"] / ms_step, 3) if (sharded and g and g.get("n1")) else None),
            "comm_share_of_proof": (round(sum(g["group"]["comm_ms_per_proof_rank0"].values()) / ms_step, 3) if (sharded and g and g.get("group")) else None),
            "vs_baseline": None, "dtype": "u32 (M31 / QM31 modular arithmetic)",
            "data": "fib19.bf execution trace (199246 VM steps), synthetic in the sense of the contract: a bundled program, no external data",
            "parity_checked": parity_checked,
            "parity": {"proof_sha256": digest, "proof_bytes": len(proof), "expected_sha256": want["sha256"] if want else None,
                       "expected_from": "tests/golden/fib19_lmr24_oracle_proof.json (CPU oracle's proof of this workload under the same conventions)" if want else None,
                       "conventions": list(conv), "own_verifier_accepts": bool(verified)},
            "config": {"workload": ("fib19.bf (BASELINE config 2; largest component 2^20 table rows = 2^24 domain rows, Blake2s Merkle), 1 proof per step"
                                    + (f" = ONE proof over the {world}-GPU shard group (BASELINE config 4's size: a 2^24-domain-row trace, column-sharded transforms, row-sharded "
                                       "Merkle / constraints / quotients / folds, RCCL); the synthetic 2^24-row trace of configs 3/4 and config 5 ride in strong_scaling" if sharded else "")
                                    + "; the metric's own point ('at 2^22 rows': synthetic nested-counter trace, 2^22 domain rows, LOG_MAX_ROWS 22) is config.metric_point"),
                       "log_max_rows": args.log_max_rows, "cells_per_proof": trace.cells, "main_cells": trace.main_cells, "interaction_cells": trace.interaction_cells,
                       "component_log_sizes": trace.log_sizes,
                       "parallelism": ("shard group: one proof over all ranks (column-sharded transforms, row-sharded Merkle/constraints/quotients/folds, RCCL)" if sharded else "replicas") if world > 1 else "single",
                       "ranks_started_by": ("bench.py itself (no launcher)" if os.environ.get("BFHIP_BENCH_SELF_LAUNCHED") else "the launcher") if world > 1 else None,
                       "proofs_in_flight_per_gpu": args.inflight, "host_thread_pinned_to_cpu": pinned_host_thread.cpu,
                       "preprocessed_tree": "reused across proofs" if args.reuse_preprocessed else "recommitted every proof (as the reference)",
                       "proof_bytes": len(proof), "phase_ms_last_step": {k: round(v * 1e3, 2) for k, v in phases.items()},
                       # ---- the metric's own numbers, inside the object the driver's record keeps whole ----
                       "metric_point": ({k: metric_point[k] for k in ("rows", "ms_per_proof", "cells_per_s", "sha256", "verified")} if metric_point else None),
                       "batch": batch_summary(pipelined),
                       "sweep_ms": ({f"2^{p['log_domain_rows']}": p["ms_per_proof"] for p in sweep} if isinstance(sweep, list) else None),
                       "headline_2^22": ({"cells_per_s": p22["cells_per_s"], "ms_per_proof": p22["ms_per_proof"], "cells": p22["cells"], "workload": p22_name,
                                          **{k2: p22[k2] for k2 in ("roofline", "gpu_busy_frac_estimate", "sum_of_kernel_ms", "kernels_ms_per_proof_instrumented") if k2 in p22}} if p22 else None)},
            "roofline": S["roofline"], "fft": S["fft"], "pipelined": pipelined, "sweep": sweep, "poseidon252": S["poseidon"],
        }
        if world > 1 and g:
            if g.get("shard_error"):
                out["shard_group_error"] = g["shard_error"] + " — value / ms_per_step are the REPLICAS (weak scaling) instead"
            if sharded:
                head = {"ms_per_proof": round(ms_step, 3), "n1_ms_per_proof": round(g["n1"]["ms_per_proof"], 3), "speedup_vs_n1": out["speedup_vs_n1"], "n1_what": g["n1"]["note"],
                        "comm_share_of_proof": out["comm_share_of_proof"], "identical_to_n1": g["n1"]["proof_sha256"] == digest, "cells_per_s": out["value"], "proof_sha256": digest,
                        "parity_checked": parity_checked, **g["group"]}
                rows = {"fib19": head}
                for nm, st in g["extra_stages"].items():
                    rows[nm] = ({k: st[k] for k in ("ms_per_proof", "n1_ms_per_proof", "speedup_vs_n1", "comm_share_of_proof", "identical_to_n1", "cells_per_s", "proof_sha256", "verified",
                                                   "comm_ms_per_proof_rank0", "error", "n1_error", "in_flight", "what", "proofs_timed", "identical_to_the_headline_proof",
                                                   "ms_per_proof_one_in_flight", "gain_vs_one_in_flight") if k in st} if isinstance(st, dict) else {"error": st})
                    if "ms_per_proof" in rows[nm] and "in_flight" in rows[nm]:
                        rows[nm]["speedup_vs_n1"] = round(g["n1"]["ms_per_proof"] / rows[nm]["ms_per_proof"], 3)
                if extras_error:
                    rows["extra_stages_error"] = extras_error
                out["strong_scaling"] = {"n_gpus": world, "what": "ONE proof over all N GPUs (shard group, RCCL, one process per GPU); n1 = the same proof by rank 0 alone on its GPU, the other ranks idle, timed in this run",
                                         "transport": g["group"]["transport"], "workloads": rows}
                out["replicas"] = g["replica_line"]
                out["replicas_pool"] = g.get("replicas_pool")      # node throughput at the metric's size: a pool of 3 per GPU, one caller thread per rank
            if shard_probe_result is not None:
                # the same stages by ONE process driving all N GPUs over the in-process transport (peer copies), and on request the RCCL child probes
                out["shard_group_single_process"] = shard_probe_result.get("single_process")
                if args.rccl_child_probe:
                    out["shard_group_child_probe"] = {k: v for k, v in shard_probe_result.items() if k != "single_process"}
        if S["cpu_baseline"] is not None:
            out["cpu_baseline"] = S["cpu_baseline"]
        return out

    def roofline_of(rep, sharded_world):
        return None if rep is None else dominant_roofline(rep, args.steps, sharded_world)

    # ---- the timed region -------------------------------------------------------------------------------------------------------------------------------------
    g = None
    sharded = dist is not None and world > 1 and not args.replicas      # N > 1: ONE proof over all ranks is the headline (strong scaling)
    if sharded and args.inflight > 1:
        raise SystemExit("a shard group proves one trace at a time: --inflight needs --replicas (or one GPU)")
    if sharded:
        def emit_partial(gp, extras_error):
            # second watchdog of an N > 1 run: the extra stages never came back, the headline group's numbers stand
            S.update(sharded=True, dt=gp["dt"], proof=gp["proof"], phases=gp["phases"], total_cells=trace.cells, roofline=roofline_of(gp.get("group_rep"), world))
            print(json.dumps(build_line(S, gp, extras_error)), flush=True)
        g = group_mod.run(args, R, pkg, replicas, ctx, trace, device, conv, one_step, sync, start_events, pin, profile_report, emit_partial)
        sharded = g["sharded"]
    if sharded:
        dt, proof, phases, total_cells = g["dt"], g["proof"], g["phases"], trace.cells      # all ranks proved the same one
    else:
        # replicas as the headline: N = 1, --replicas, or the group failed (the contract: W untimed, barrier + sync, K timed, barrier + sync, MAX over ranks)
        R.note("replicas (headline)")
        with pin():
            dt, (proof, phases) = replicas.timed_region(one_step, args.steps, args.warmup, dist=dist, sync_fn=sync, backend_tensor=R.cuda_t, on_timed_start=start_events)
        total_cells = replicas.aggregate_units(trace.cells * args.inflight, dist=dist, backend_tensor=R.cuda_t)
    S.update(sharded=sharded, dt=dt, proof=proof, phases=phases, total_cells=total_cells)

    # ---- roofline of the dominant kernel over the TIMED region; the sustained clock right behind it; the FFT kernels from one extra untimed proof -------------
    if not args.no_kernel_events:
        rep = g["group_rep"] if (sharded and g.get("group_rep") is not None) else profile_report(lib, ctx)
        lib.bfhip_profile_enable(ctx._h, 0)
        S["roofline"] = roofline_of(rep, world if sharded else 0)
        if not args.no_clock_probe:
            add_sustained_clock(S["roofline"], ctx)
        if world == 1:
            lib.bfhip_profile_enable(ctx._h, 1)
            lib.bfhip_profile_reset(ctx._h)
            trace.prove(args.log_max_rows)
            full = profile_report(lib, ctx)
            lib.bfhip_profile_enable(ctx._h, 0)
            S["fft"] = fft_report(full)
            if S["fft"]:
                S["roofline"]["all_kernels_ms_per_proof_instrumented"] = {k: round(v["total_ms"], 3) for k, v in sorted(full.items(), key=lambda kv: -kv[1]["total_ms"])}
                # every Blake2s compression of the proof's trees, whichever kernel ran it (fixed by the protocol)
                S["roofline"]["compressions_per_proof_all_merkle_kernels"] = round(sum(v.get("units", 0) for k, v in full.items() if k in ("k_merkle_layer", "k_merkle_subtree", "k_merkle_top", "k_fri_tail", "k_fri_layer")))

    if world == 1 and not args.no_sweep and rank == 0:
        try:
            with pin():
                S["sweep"] = run_sweep(pkg, device, args.sweep_steps, [int(v) for v in args.sweep_logs.split(",")])
        except Exception as e:      # the sweep must never cost the headline line
            S["sweep"] = {"error": repr(e)}
        if not args.no_poseidon:
            try:
                with pin():
                    S["poseidon"] = run_poseidon_point(pkg, device, args.poseidon_log)
            except Exception as e:
                S["poseidon"] = {"error": repr(e)}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            S["cpu_baseline"] = cpu_baseline_block(args.cpu_baseline, trace.cells, total_cells * args.steps / dt, hashlib.sha256(proof).hexdigest(), dt / args.steps, trace.log_sizes, args.log_max_rows)
        print(json.dumps(build_line(S, g)), flush=True)
    if args.reuse_preprocessed:
        lib.bfhip_ctx_reuse_preprocessed(ctx._h, 0)
    for c2, t2 in extra:
        t2.close(); c2.close()
    trace.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
