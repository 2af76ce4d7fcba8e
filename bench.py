#!/usr/bin/env python3
"""Benchmark of the hot path: prove_brainfuck on MI355X, metric = trace cells committed+proved per second (BASELINE.json).

A "step" is one complete proof (preprocessed commitment .. decommitment, crates/brainfuck_prover/src/brainfuck_air/mod.rs:493-734)
of one trace whose row-granular table columns are already resident in HBM. Workload at N=1: BASELINE.json configs[1]
(fib19.bf, largest component 2^20 table rows = 2^24 domain rows, Blake2s Merkle, LOG_MAX_ROWS = 24).
N > 1: one process per GPU. Default: every rank proves its own independent trace (replicas, weak scaling, no data-path collective).
--shard: the N ranks prove ONE trace together (shard group, strong scaling).

Prints ONE JSON line on rank 0 (driver contract). Checked and measured inside this run:
  parity_checked  SHA-256 of the last timed proof == the committed digest of the CPU oracle's proof of the same workload
  roofline        dominant kernel, HIP events on the library's own stream over the timed region (k_merkle_layer: integer-VALU bound,
                  Blake2s compressions counted from the launch shapes; HBM figure beside it)
  fft             the circle-FFT kernels' moved and algorithmic GB/s (north-star figure), from one extra untimed, fully instrumented proof
  sweep           synthetic nested-counter traces of 2^20..2^26 domain rows (BASELINE metric "at 2^22 rows": config.headline_2^22)
  poseidon252     BASELINE config 5 on one GPU: the 2^26-row synthetic trace with the Poseidon252 MerkleChannel variant
  shard_group     N > 1 only: ONE proof over all N GPUs (strong scaling, RCCL), taken in child processes before the timed replicas run: the
                  bench workload, a 2^24-row trace (configs 3/4) and the 2^26-row Poseidon252 trace (config 5); SHA-256 of each proof
  cpu_baseline    the CPU oracle ("port") — see cpu_baseline()
"""
import argparse
import ctypes
import hashlib
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

# Integer-VALU roofline of the Blake2s kernel: one compression = 977 VALU lane-ops in the compiled kernel (v_add3_u32 / v_xor_b32 /
# v_alignbit_b32; llvm-objdump of k_merkle_layer), and the chip retires 256 CU x 4 SIMD x 16 int lanes/clk x 2.4 GHz = 39.3 T such
# lane-ops/s (half the fp32-FMA issue rate; tools/ubench_blake.hip measures 39.9 G compressions/s = 39.0 T lane-ops/s in registers).
VALU_OPS_PER_COMPRESSION = 977
VALU_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12

FIB19 = "+++++++++++++++++>+>+<<[->>[->+>+<<]<[->>+<<]>>[-<+>]>[-<<<+>>>]<<<<]>>."  # tests/golden/programs/fib19.bf (workload input)

# Synthetic padded traces: "+"*a "[>" "+"*b "[>+<-]<-]" — the Memory component lands exactly on
# 2^k domain rows for (a, b) = (14, 250 * 2^(k-20)); proved with LOG_MAX_ROWS = k.
SWEEP = {k: (14, 250 << (k - 20)) for k in range(20, 27)}


def sweep_program(k):
    a, b = SWEEP[k]
    return "+" * a + "[>" + "+" * b + "[>+<-]<-]"


def load_package():
    name = "stwo_brainfuck_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "stwo-brainfuck_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def kernel_sources_sha256():
    """SHA-256 over the sources of the dominant kernel (what a committed counter file must have been measured on)."""
    h = hashlib.sha256()
    for rel in ("stwo-brainfuck_amd/csrc/merkle.hip", "stwo-brainfuck_amd/csrc/kernels.h", "stwo-brainfuck_amd/csrc/m31.h"):
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()


def committed_digests():
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "fib19_lmr24_oracle_proof.json")))
    except Exception:
        return {}


def cpu_baseline(cells_per_proof, full=False):
    """CPU baseline, kind "port": the CPU oracle (the Rust reference and its stwo dependency cannot be built on this image).

    value = the port on THE BENCH WORKLOAD ITSELF (fib19.bf, LOG_MAX_ROWS 24). On a host with >= 32 cores (the GPU boxes: 33 s with 64
    OpenMP threads) the proof is timed LIVE in this run (`full`) and its SHA-256 is reported, so the baseline is like for like: same
    workload, same box, same bytes. On a small host (the 8-core build container needs ~9 minutes) the value is the committed measurement taken
    when the parity digest was generated (tests/golden/fib19_lmr24_oracle_proof.json) and `live` holds a bounded sample timed on this box
    (collatz.bf at LOG_MAX_ROWS 21: a 13x smaller trace, on which the port's cells/s is much higher — not comparable with `value`).
    The port is a scalar restatement with OpenMP loops, not a stand-in for SimdBackend + rayon: the north-star target ">= 10x the
    reference's parallel CPU prover" is UNDETERMINED here, whatever this ratio says."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import Oracle
    orc = Oracle()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    out = {"unit": "trace cells/s", "kind": "port",
           "vs_reference_parallel_cpu": "undetermined: the Rust reference (SimdBackend + rayon) cannot be built here; the port is a scalar OpenMP restatement"}
    if full:
        threads = min(avail, 64)
        orc.L.orc_set_threads(threads)
        t0 = time.time()
        proof, _, _ = orc.prove(FIB19, b"", log_max_rows=24)
        sec = time.time() - t0
        out.update({"value": cells_per_proof / sec, "cores": threads, "host_cores_available": avail,
                    "sample": f"fib19.bf at LOG_MAX_ROWS 24 (the bench workload itself, {cells_per_proof} cells), one proof timed live on this box: {sec:.1f} s with {threads} OpenMP threads",
                    "proof_sha256": hashlib.sha256(proof).hexdigest()})
        return out
    fx = committed_digests().get("stwo")
    # bounded live sample on this box's cores
    code = open(os.path.join(ROOT, "tests", "golden", "programs", "collatz.bf")).read()
    log_sizes, steps = orc.log_sizes(code, b"7\n")
    main_cols = [8, 8, 4, 9, 13, 13, 11, 11, 11, 11, 11, 11, 7]
    inter_cols = [4, 4, 4, 12, 4, 4, 4, 4, 4, 4, 4, 4, 4]
    cells = sum((m + i) << l for m, i, l in zip(main_cols, inter_cols, log_sizes))
    best_sec, best_threads, tried = None, 1, []
    for threads in sorted({min(avail, t) for t in (16, 32, 64)}):   # the port's OpenMP loops stop scaling at a few dozen threads
        orc.L.orc_set_threads(threads)
        _, _, sec = orc.prove(code, b"7\n", log_max_rows=max(log_sizes))
        tried.append(f"{threads}t {sec:.1f}s")
        if best_sec is None or sec < best_sec:
            best_sec, best_threads = sec, threads
    live = {"value": cells / best_sec, "cores": best_threads, "host_cores_available": avail,
            "sample": f"collatz.bf input '7\\n' ({steps} VM steps, {cells} cells, LOG_MAX_ROWS={max(log_sizes)}), one proof per OpenMP team size ({', '.join(tried)}), best reported"}
    if fx:
        out.update({"value": cells_per_proof / fx["oracle_seconds"], "cores": 8,
                    "sample": f"fib19.bf at LOG_MAX_ROWS 24 — the bench workload itself ({cells_per_proof} cells): {fx['oracle_seconds']} s for one proof, measured once on the 8 cores of "
                              "the build container when the parity digest was generated (tests/golden/fib19_lmr24_oracle_proof.json); NOT timed in this run (use --cpu-baseline full)",
                    "live": live})
    else:
        out.update(live)
    return out


MAIN_COLS = [8, 8, 4, 9, 13, 13, 11, 11, 11, 11, 11, 11, 7]      # TraceColumn::count().0 per component, claim order (mod.rs:85-99)
LOGUP_COLS = [1, 1, 1, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1]


def simdbackend_work_counts(log_sizes, lmr):
    """Blake2s compressions and radix-2 butterflies a SimdBackend-shaped prover performs for one proof of a trace with these component sizes:
    every column FULL SIZE (the reference broadcasts each table row into 16 lanes and its backend does not know it: memory/table.rs:95-104),
    mixed-degree Merkle trees with one compression per 64 message bytes (children 64 B, then 16 column words per block), interpolate +
    evaluate-on-the-blowup-domain per committed column (mod.rs:497,550-583,690-723 and the composition commit inside prover::prove). Only these
    two loops are counted — a lower bound of the work."""
    def tree(col_logs):
        mx, total = max(col_logs), 0
        for lg in range(mx, -1, -1):
            ncols = sum(1 for c in col_logs if c == lg)
            msg = (64 if lg < mx else 0) + 4 * ncols
            total += (1 << lg) * max(1, -(-msg // 64))
        return total
    pre = [l + 1 for l in range(lmr, 3, -1)]
    main = [l + 1 for l, m in zip(log_sizes, MAIN_COLS) for _ in range(m)]
    inter = [l + 1 for l, n in zip(log_sizes, LOGUP_COLS) for _ in range(4 * n)]
    comp_log = max(log_sizes) + 1
    comp = [comp_log + 1] * 4
    sizes = sorted(set(pre + main + inter + comp), reverse=True)
    trees = [pre, main, inter, comp, [sz for sz in sizes for _ in range(4)]] + [[line] * 4 for line in range(sizes[0] - 1, 1, -1)]
    compressions = sum(tree(t) for t in trees)
    butterflies = 0
    for lde in pre + main + inter + comp:      # iFFT on 2^(lde-1) points, FFT on 2^lde points: n/2 butterflies per layer
        n = lde - 1
        butterflies += n * (1 << (n - 1)) + lde * (1 << (lde - 1))
    return compressions, butterflies


def simd_bound(gpu_seconds_per_proof, log_sizes, lmr, seconds_each=4.0):
    """cpu_baseline.simd_bound: the host's vector units on the two loops the reference's SimdBackend + rayon prover cannot avoid (oracle/
    simd_bound.cpp: 16-lane Blake2s compression, packed M31 butterfly; AVX-512 if the host has it, else AVX2; every hardware thread busy,
    operands in registers) -> a LOWER bound of the reference's proving time on this host and the speedup the GPU has over that bound."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import Oracle
    L = Oracle().L
    L.orc_simd_bound.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    out = (ctypes.c_double * 4)()
    L.orc_simd_bound(avail, seconds_each, out)
    comp_rate, bfly_rate, width, threads = out[0], out[1], int(out[2]), int(out[3])
    if width == 0 or comp_rate <= 0 or bfly_rate <= 0:
        return {"error": "the host has neither AVX-512 nor AVX2"}
    # One thread alone: what a core of this host does when nothing else competes. The all-thread run above is what the box GIVES this process
    # (cgroup CPU quotas and the other tenants of the host included); `threads x single-thread rate` is what the hardware could do at most
    # (SMT siblings do not double a vector loop, so it overstates the host) — the stricter of the two bounds decides the north-star sentence.
    one = (ctypes.c_double * 4)()
    L.orc_simd_bound(1, min(seconds_each, 2.0), one)
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().split()
        cpu_quota = None if quota[0] == "max" else round(int(quota[0]) / int(quota[1]), 1)
    except Exception:
        cpu_quota = None
    try:
        sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        smt = max(1, len([x for part in sib.split(",") for x in ([part] if "-" not in part else range(int(part.split("-")[0]), int(part.split("-")[1]) + 1))]))
    except Exception:
        smt = 1
    physical = max(1, avail // smt)
    comps, bflies = simdbackend_work_counts(log_sizes, lmr)
    t_hash, t_fft = comps / comp_rate, bflies / bfly_rate
    # whole host: every PHYSICAL core at the rate one thread reaches alone (SMT siblings share the vector ports), or the all-thread run if faster
    ideal_comp, ideal_bfly = max(comp_rate, one[0] * physical), max(bfly_rate, one[1] * physical)
    t_ideal = comps / ideal_comp + bflies / ideal_bfly
    ratio_measured = (t_hash + t_fft) / gpu_seconds_per_proof
    ratio = t_ideal / gpu_seconds_per_proof
    granted = ("the %s cores the box's CPU quota grants this process" % cpu_quota) if cpu_quota else "all %d hardware threads" % threads
    return {"instruction_set": "AVX-512 (16 x u32 per register)" if width == 512 else "AVX2 (two 8-lane halves per 16 lanes)", "threads": threads, "host_cores_available": avail,
            "physical_cores": physical, "smt_threads_per_core": smt,
            "blake2s_compressions_per_s": comp_rate, "m31_butterflies_per_s": bfly_rate,
            "single_thread": {"blake2s_compressions_per_s": one[0], "m31_butterflies_per_s": one[1]}, "cgroup_cpu_quota_cores": cpu_quota,
            "seconds_lower_bound_whole_host": t_ideal,
            "gpu_over_simd_bound_as_measured_on_all_threads": round(ratio_measured, 2),
            "work_counted": {"blake2s_compressions": comps, "m31_butterflies": bflies,
                             "note": "full-size columns (the reference's SimdBackend does not exploit the 16x lane broadcast), Merkle + channel hashing and the column transforms only"},
            "seconds_lower_bound": {"hashing": t_hash, "transforms": t_fft, "total": t_hash + t_fft},
            "cells_per_s_upper_bound": None,
            "gpu_over_simd_bound": round(ratio, 2),       # against the STRICTER bound (physical cores x single-thread rate, or the all-thread run if faster)
            "north_star_10x": {
                "on_the_cpu_this_box_grants": ("%s: the GPU proof is %.1fx faster than the fastest the vector units could hash and transform this trace as run on %s"
                                               % ("met" if ratio_measured >= 10.0 else "not determined by the bound", ratio_measured, granted)),
                "on_the_whole_host": ("%s: against %d physical cores each at the rate one thread reaches alone (registers only, perfect scaling, no memory traffic) the GPU proof is %.1fx faster; "
                                      "the real reference (constraints, quotients, logUp, memory traffic, rayon) is slower than this bound by an unknown factor"
                                      % ("met" if ratio >= 10.0 else "not determined by the bound", physical, ratio))},
            "stands_in_for": "brainfuck_prover prove --features parallel (README.md:23-36), 'Proof generation time' (bin/brainfuck_prover.rs:137-139): not buildable here"}


def pick_device(local_rank, n_visible, override=None):
    """One process per GPU: rank r drives device LOCAL_RANK. A launcher that narrows each rank's view to its own GPU
    (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank) leaves one visible device, numbered 0, on every rank."""
    if override is not None:
        return override
    return local_rank if local_rank < n_visible else local_rank % max(n_visible, 1)


def profile_report(lib, ctx):
    js = ctypes.c_void_p()
    lib.bfhip_profile_report(ctx._h, ctypes.byref(js))
    rep = json.loads(ctypes.string_at(js).decode())
    lib.bfhip_free_host(js)
    return rep


def point_roofline(pkg, c, tr, lmr, sec_per_proof):
    """The metric's own size (BASELINE 'at 2^22 rows'): kernel-time split, GPU-busy fraction and the dominant kernel's roofline of that proof, from
    two extra untimed proofs — one with the dominant kernel bracketed by HIP events per run of launches (as in the timed region of the main
    workload), one with every kernel bracketed (time split; the event pairs themselves stretch small proofs, so the busy fraction is the sum of
    the kernel times over the UN-instrumented wall time and is an upper estimate when streams overlap)."""
    lib = pkg.lib()
    lib.bfhip_profile_enable(c._h, 2); lib.bfhip_profile_reset(c._h)
    tr.prove(lmr, want_json=False); c.sync()
    dom = profile_report(lib, c)
    lib.bfhip_profile_enable(c._h, 1); lib.bfhip_profile_reset(c._h)
    tr.prove(lmr, want_json=False); c.sync()
    full = profile_report(lib, c)
    lib.bfhip_profile_enable(c._h, 0)
    out = {"kernels_ms_per_proof_instrumented": {k: round(v["total_ms"], 3) for k, v in sorted(full.items(), key=lambda kv: -kv[1]["total_ms"])}}
    tot = sum(v["total_ms"] for v in full.values())
    out["sum_of_kernel_ms"] = round(tot, 3)
    out["gpu_busy_frac_estimate"] = round(min(1.0, tot / (sec_per_proof * 1e3)), 3)
    d = dom.get("k_merkle_layer")
    if d and d.get("units", 0) > 0 and d["total_ms"] > 0:
        tops = d["units"] * VALU_OPS_PER_COMPRESSION / (d["total_ms"] * 1e-3) / 1e12
        out["roofline"] = {"kernel": "k_merkle_layer", "bound": "valu", "achieved": round(tops, 2), "peak": round(VALU_PEAK_TOPS, 2), "unit": "Tops/s (int32 VALU lane-ops)",
                           "frac": round(tops / VALU_PEAK_TOPS, 4), "launches": d["calls"], "avg_launch_us": round(d["total_ms"] / d["calls"] * 1e3, 2),
                           "compressions_per_proof": round(d["units"]), "kernel_ms_per_proof": round(d["total_ms"], 3),
                           "share_of_proof": round(d["total_ms"] / (sec_per_proof * 1e3), 3),
                           "hbm": {"achieved": round(d["bytes"] / d["total_ms"] / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(d["bytes"] / d["total_ms"] / 1e6 / HBM_PEAK_GBS, 4)}}
    return out


def run_sweep(pkg, device, steps, logs):
    """Synthetic padded traces of 2^k domain rows, k in `logs`: one context sized for the largest, LOG_MAX_ROWS = k per point."""
    out = []
    c = pkg.Context(device, max_log_domain=max(logs) + 2)
    try:
        for k in logs:
            tr = pkg.Trace(c, sweep_program(k), b"")
            try:
                assert max(tr.log_sizes) == k, (k, tr.log_sizes)
                proof, _ = tr.prove(k)            # warm-up (the arena grows on the first proof of a size)
                c.sync()
                t0 = time.perf_counter()
                for _ in range(steps):
                    proof, _ = tr.prove(k)
                c.sync()
                dt = (time.perf_counter() - t0) / steps
                ok, why = pkg.verify_brainfuck(proof, k)
                row = {"log_domain_rows": k, "log_max_rows": k, "vm_steps": tr.n_steps, "cells": tr.cells, "ms_per_proof": round(dt * 1e3, 3),
                       "cells_per_s": tr.cells / dt, "proof_bytes": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(), "verified": bool(ok)}
                if k == 22:
                    row.update(point_roofline(pkg, c, tr, k, dt))
                out.append(row)
            finally:
                tr.close()
    finally:
        c.close()
    return out


def run_poseidon_point(pkg, device, log):
    """BASELINE config 5 on one GPU: the synthetic 2^log-row trace proved with the Poseidon252 MerkleChannel variant (one warm-up, one timed
    proof; the shard probe proves the same trace over N GPUs and reports the same SHA-256)."""
    conv = (0, 0, 0, 1)
    c = pkg.Context(device, max_log_domain=log + 2)
    try:
        c.set_conventions(*conv)
        tr = pkg.Trace(c, sweep_program(log), b"")
        try:
            tr.prove(log)
            c.sync()
            t0 = time.perf_counter()
            proof, phases = tr.prove(log)
            c.sync()
            dt = time.perf_counter() - t0
            ok, _ = pkg.verify_brainfuck(proof, log, conv)
            return {"log_domain_rows": log, "log_max_rows": log, "conventions": list(conv), "cells": tr.cells, "ms_per_proof": round(dt * 1e3, 1),
                    "cells_per_s": tr.cells / dt, "proof_bytes": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(), "verified": bool(ok),
                    "phase_ms": {k: round(v * 1e3, 1) for k, v in phases.items()}}
        finally:
            tr.close()
    finally:
        c.close()


def probe_stages(args):
    """(name, program, LOG_MAX_ROWS, conventions, warm-up proofs, timed proofs, overlap mask): the bench workload, then BASELINE configs 3/4 (a
    2^24-row synthetic trace) and 5 (a 2^26-row trace with the Poseidon252 MerkleChannel) proved by the whole group, and last the bench workload
    again with the column -> row exchanges on the partner stream (bfhip_ctx_set_overlap bit 2: unmeasured on hardware until a multi-GPU run)."""
    conv = tuple((([int(v) for v in args.conventions.split(",")]) + [0, 0, 0, 0])[:4])
    stages = [("fib19", FIB19, args.log_max_rows, conv, 2, args.probe_steps, 0)]
    if not args.probe_fib19_only:
        stages.append(("trace_2p24_blake2s", sweep_program(24), 24, (0, 0, 0, 0), 1, 3, 0))
        stages.append(("trace_2p26_poseidon252", sweep_program(26), 26, (0, 0, 0, 1), 1, 1, 0))
        stages.append(("fib19_exchange_overlap", FIB19, args.log_max_rows, conv, 1, args.probe_steps, 4))
    return stages


def probe_n1_reference(pkg, device, code, lmr, conv, max_log, steps):
    """The same workload proved by ONE GPU alone (a context of its own, outside the group), timed right before the group proves it: what
    `speedup_vs_n1` divides by. Every rank does this on its own GPU at the same time, so it costs the probe one proof's time, not N."""
    c = pkg.Context(device, max_log_domain=max_log)
    try:
        c.set_conventions(*conv)
        tr = pkg.Trace(c, code, b"")
        try:
            tr.prove(lmr); c.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                proof, _ = tr.prove(lmr)
            c.sync()
            return (time.perf_counter() - t0) / steps, hashlib.sha256(proof).hexdigest()
        finally:
            tr.close()
    finally:
        c.close()


def probe_run_stages(pkg, members, stages, out, flush, is_rank0, ref_device=None, max_log=26):
    """Runs every stage on `members` (the contexts this process drives: one with RCCL, all N of an in-process group — one host thread
    each). Results go to out["stages"][name]; a failed stage ends the probe (the other members may be inside its collectives).
    ref_device: the GPU this process times the one-GPU reference of every stage on (None: no reference)."""
    import threading
    for name, code, lmr, conv, warm, steps, overlap in stages:
        row = {"log_max_rows": lmr, "conventions": list(conv), "overlap_mask": overlap}
        out["stages"][name] = row
        if ref_device is not None and not name.endswith("_exchange_overlap"):
            try:
                n1_sec, n1_sha = probe_n1_reference(pkg, ref_device, code, lmr, conv, max_log, max(1, min(steps, 3)))
                row.update({"n1_ms_per_proof": round(n1_sec * 1e3, 3), "n1_proof_sha256": n1_sha})
            except Exception as e:
                row["n1_error"] = repr(e)
            flush()
        n = len(members)
        gate = threading.Barrier(n)
        res, errors = [None] * n, []

        def run(k):
            ctx, trace = members[k], None
            try:
                ctx.set_conventions(*conv)
                ctx.set_overlap(overlap)
                trace = pkg.Trace(ctx, code, b"")
                before = ctx.group_stats()
                for _ in range(warm):
                    trace.prove(lmr)
                ctx.sync()
                t_before = ctx.group_times()
                gate.wait(timeout=600)
                t0 = time.perf_counter()
                for _ in range(steps):
                    proof, phases = trace.prove(lmr)
                ctx.sync()
                dt_k = time.perf_counter() - t0
                t_after = ctx.group_times()
                res[k] = (dt_k, proof, phases, trace.cells, before, ctx.group_stats(), {kk: (t_after[kk] - t_before[kk]) / steps for kk in t_after})
            except Exception as e:
                errors.append(repr(e))
                gate.abort()
            finally:
                if trace is not None:
                    trace.close()

        threads = [threading.Thread(target=run, args=(k,)) for k in range(n)]
        [t.start() for t in threads]; [t.join() for t in threads]
        if errors:
            row["error"] = "; ".join(errors)
            flush()
            raise RuntimeError(row["error"])
        dt = max(r[0] for r in res) / steps
        _, proof, phases, cells, before, after, comm_ms = res[0]
        row.update({"ms_per_proof": round(dt * 1e3, 3), "cells": cells, "cells_per_s": cells / dt, "steps": steps,
                    "proof_bytes": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(),
                    "all_members_same_proof": all(r[1] == proof for r in res),
                    "phase_ms_last_proof": {k: round(v * 1e3, 2) for k, v in phases.items()},
                    "per_proof": {k: round((after[k] - before[k]) / (warm + steps), 1) for k in after},
                    # where a proof over several GPUs spends its time: GPU-side milliseconds inside the collectives (HIP-event pairs on the rank's
                    # stream: includes waiting for the slowest peer), rank 0 and the maximum over the ranks; the rest of ms_per_proof is compute
                    "comm_ms_per_proof_rank0": {k: round(v, 3) for k, v in comm_ms.items()},
                    "comm_ms_per_proof_max_rank": {k: round(max(r[6][k] for r in res), 3) for k in comm_ms},
                    "comm_share_of_proof": round(sum(comm_ms.values()) / (dt * 1e3), 3)})
        if "n1_ms_per_proof" in row:
            row["speedup_vs_n1"] = round(row["n1_ms_per_proof"] / row["ms_per_proof"], 3)
            row["identical_to_n1"] = row["n1_proof_sha256"] == row["proof_sha256"]
        if is_rank0:
            row["verified"] = bool(pkg.verify_brainfuck(proof, lmr, conv)[0])
        if name.startswith("fib19"):
            want = next((d for d in committed_digests().values() if tuple(d.get("conventions", ())) == conv and d.get("log_max_rows") == lmr), None)
            row["parity_checked"] = bool(want is not None and row["proof_sha256"] == want["sha256"])
        flush()


def strong_scaling_summary(probe, world):
    """Top-level digest of the shard probe for a SCALE record: per workload {ms_per_proof with all N GPUs on ONE proof, the one-GPU time of the
    same proof measured in the same run, speedup_vs_n1, comm_share_of_proof, identical_to_n1}; RCCL (one process per GPU) first, the
    in-process transport (one process driving all GPUs) beside it."""
    def digest(p):
        rows = {}
        for name, st in (p or {}).get("stages", {}).items():
            if "ms_per_proof" not in st:
                rows[name] = {"error": st.get("error", "stage did not complete")}
                continue
            rows[name] = {k: st[k] for k in ("ms_per_proof", "n1_ms_per_proof", "speedup_vs_n1", "comm_share_of_proof", "identical_to_n1", "all_members_same_proof",
                                             "cells_per_s", "proof_sha256") if k in st}
        return {"transport": (p or {}).get("transport"), "error": (p or {}).get("error"), "workloads": rows}
    return {"n_gpus": world, "what": "ONE proof over all N GPUs (shard group); value / ms_per_step above are the replicas (N independent proofs)",
            "rccl": digest(probe), "single_process": digest(probe.get("single_process")) if isinstance(probe, dict) and "single_process" in probe else None}


def shard_probe(args):
    """Child-process mode (--shard-probe): ONE proof over all N GPUs, a few proofs per stage (probe_stages). Two transports:
    default — this rank's child joins the other ranks' children in an RCCL shard group (the 128-byte unique id travels through a file);
    --probe-local — rank 0's child alone drives all N GPUs from N host threads over the library's in-process transport (peer copies).
    No torch, no torch.distributed: the parent keeps its process group for the contract's timing protocol. The result file is rewritten
    after every stage, so a stage that hangs (the parent kills this child on its timeout) does not cost the earlier ones."""
    rank, world, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    out = {"n_gpus": world, "rank": rank, "stages": {}}

    def flush():
        with open(args.probe_out + ".tmp", "w") as f:
            json.dump(out, f)
        os.replace(args.probe_out + ".tmp", args.probe_out)

    try:
        pkg = load_package()
        stages = probe_stages(args)
        max_log = max(s[2] for s in stages) + 2
        if args.probe_local:
            n_dev = pkg.device_count()
            devices = [pick_device(r, n_dev, args.device) for r in range(world)]
            out["devices"] = devices
            members = [pkg.Context(d, max_log_domain=max_log) for d in devices]
            group = pkg.LocalGroup(world) if world > 1 else None
            if group is not None:
                import threading
                errs = []
                def join(r):
                    try:
                        members[r].join_local_group(group, r)
                    except Exception as e:
                        errs.append(repr(e))
                th = [threading.Thread(target=join, args=(r,)) for r in range(world)]
                [t.start() for t in th]; [t.join() for t in th]
                if errs:
                    raise RuntimeError("; ".join(errs))
        else:
            members = [pkg.Context(pick_device(local_rank, pkg.device_count(), args.device), max_log_domain=max_log)]
            if world > 1:                       # world == 1: the stages on a single GPU (how the probe itself is tested on a 1-GPU box)
                idf = args.probe_id_file
                if rank == 0:
                    with open(idf + ".tmp", "wb") as f:
                        f.write(pkg.rccl_unique_id())
                    os.replace(idf + ".tmp", idf)
                t0 = time.time()
                while not os.path.exists(idf):
                    if time.time() - t0 > 60:
                        raise RuntimeError("unique id file did not appear")
                    time.sleep(0.02)
                members[0].join_rccl_group(open(idf, "rb").read(), rank, world)
        flush()
        ref_device = (devices[0] if args.probe_local else pick_device(local_rank, pkg.device_count(), args.device)) if world > 1 else None
        probe_run_stages(pkg, members, stages, out, flush, rank == 0, ref_device=ref_device, max_log=max_log)
        out["transport"] = members[0].group_info()[2]
        for m in members:
            if world > 1:
                m.leave_group()
            m.close()
    except Exception as e:
        out["error"] = repr(e)
    flush()
    return 0


def run_shard_probe(args, rank, world):
    """Parent side, BEFORE this process touches the GPU (a child must not be exec'd from a process that has initialised it).
    (1) every rank starts its RCCL probe child, waits for it (bounded) and kills exactly that PID on timeout; (2) rank 0 alone starts the
    in-process probe child (N host threads driving the N GPUs) while the other ranks wait for its completion marker.
    Returns rank 0's RCCL result with the in-process result under "single_process" (or error records)."""
    import subprocess
    import tempfile
    base = os.path.join(tempfile.gettempdir(), f"bfhip_probe_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}")
    marker = f"{base}.localdone"

    def run_child(extra, out_path):
        try:
            os.remove(out_path)
        except OSError:
            pass
        cmd = [sys.executable, os.path.abspath(__file__), "--shard-probe", "--probe-out", out_path, "--probe-id-file", f"{base}.id", "--probe-steps", str(args.probe_steps),
               "--log-max-rows", str(args.log_max_rows), "--conventions", args.conventions] + extra
        if args.probe_fib19_only:
            cmd.append("--probe-fib19-only")
        if args.device is not None:
            cmd += ["--device", str(args.device)]
        child = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        try:
            child.wait(timeout=args.probe_timeout)
        except subprocess.TimeoutExpired:
            child.kill()
            child.wait()
            try:
                partial = json.load(open(out_path))       # the stages that completed before the one that hung
            except Exception:
                partial = {"n_gpus": world}
            partial["error"] = f"probe child did not finish within {args.probe_timeout} s (killed); stages listed without ms_per_proof did not complete"
            return partial
        try:
            return json.load(open(out_path))
        except Exception as e:
            return {"n_gpus": world, "error": f"probe child left no result (exit code {child.returncode}): {e!r}"}

    if rank == 0:
        for path in (f"{base}.id", marker):
            try:
                os.remove(path)
            except OSError:
                pass
    result = run_child([], f"{base}_rank{rank}.json")
    if args.no_local_probe:
        return result
    if rank == 0:
        try:
            result["single_process"] = run_child(["--probe-local"], f"{base}_local.json")
        finally:
            open(marker, "w").close()
    else:
        t0 = time.time()
        while not os.path.exists(marker) and time.time() - t0 < args.probe_timeout + 60:
            time.sleep(0.2)
    return result


def gpu_local_cpus(device):
    """The cores next to GPU `device` (sysfs local_cpulist of its PCI function), or None when that cannot be read."""
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) != 0:
            return None
        text = open(f"/sys/bus/pci/devices/{buf.value.decode().lower()}/local_cpulist").read().strip()
        cpus = set()
        for part in text.split(","):
            if "-" in part:
                a, b = part.split("-"); cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        return cpus or None
    except Exception:
        return None


class pinned_host_thread:
    """The proving thread on ONE core next to its GPU for the duration of a timed region (restored afterwards: the CPU baseline and child
    processes use every core). A proof is ~10 Fiat-Shamir round trips with the GPU idle in each; a thread that the scheduler migrates while it
    polls adds a 0.3-0.5 ms tail to 10-15 % of the 2^22-row proofs (measured: mean 9.24 -> 9.14 ms, p90 9.50 -> 9.20 ms under taskset) — what
    any deployment does with numactl. Only a core of the GPU's own NUMA node is taken (a far core costs more than the jitter: measured); when
    the node cannot be determined nothing is pinned. Opt-in (--pin): on other boxes of the pool the same pinning changed nothing or cost 1 %."""
    cpu = None

    def __init__(self, enabled, device=0, local_rank=0, world=1):
        self.enabled, self.device, self.local_rank, self.world, self.old = enabled, device, local_rank, world, None

    def __enter__(self):
        if not self.enabled or not hasattr(os, "sched_setaffinity"):
            return self
        try:
            old = os.sched_getaffinity(0)
            near = gpu_local_cpus(self.device)
            cand = sorted(old & near) if near else []
            if not cand:
                return self
            # ranks that share a node take different cores; the first cores of a node are left to interrupt handling
            cpu = cand[(2 + 2 * self.local_rank) % len(cand)]
            os.sched_setaffinity(0, {cpu})
            self.old = old
            pinned_host_thread.cpu = cpu
        except OSError:
            self.old = None
        return self

    def __exit__(self, *exc):
        if self.old is not None:
            os.sched_setaffinity(0, self.old)
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-max-rows", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pin", action="store_true", help="pin the proving host thread to one core of the GPU's NUMA node during the timed regions (default off: the gain "
                    "is box dependent, profiles/r03_bench_host_pinning_ab.txt)")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "full", "sample"],
                    help="full: time the CPU port on the bench workload itself live on this box (~35 s with 64 threads, ~20 GB of host memory); sample: committed "
                         "full-size measurement + a bounded live sample; auto (default): full when the host has >= 32 cores and >= 48 GB of free memory")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--kernel-events", default="dominant", choices=["dominant", "all"], help="HIP-event timing of the dominant kernel only (default, ~0.5%% overhead) or of every kernel (~10%%) inside the timed region")
    ap.add_argument("--no-sweep", action="store_true", help="skip the 2^20..2^26 synthetic sweep (N=1 only; ~15 s)")
    ap.add_argument("--sweep-steps", type=int, default=3)
    ap.add_argument("--no-poseidon", action="store_true", help="skip the Poseidon252 2^26-row point (BASELINE config 5 on one GPU; ~15 s)")
    ap.add_argument("--poseidon-log", type=int, default=26)
    ap.add_argument("--sweep-logs", default="20,21,22,23,24,25,26")
    ap.add_argument("--conventions", default="0,0,0,0", help="merkle_node_hash,mix_u64,logup_mask_order,merkle_channel (include/bfhip.h bfhip_conventions); default = stwo defaults, Blake2s channel")
    ap.add_argument("--reuse-preprocessed", action="store_true", help="NOT the headline: keep the program-independent preprocessed tree across proofs (a deployment option; the reference recommits it per proof)")
    ap.add_argument("--inflight", type=int, default=1, help="proofs in flight per GPU (one host thread + HIP stream each); >1 reports pipelined throughput, no roofline")
    ap.add_argument("--shard", action="store_true", help="NOT the default: with --gpus N > 1 the N ranks prove ONE trace together (strong scaling of a single proof) "
                    "instead of N independent replicas")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) | gloo (test mode on boxes with fewer GPUs than ranks)")
    ap.add_argument("--device", type=int, default=None, help="test mode: every rank uses this device instead of LOCAL_RANK")
    ap.add_argument("--no-shard-probe", action="store_true", help="N > 1, replicas mode: skip the extra strong-scaling measurement (one proof over all N GPUs) taken in child "
                    "processes before the timed replicas run")
    ap.add_argument("--probe-steps", type=int, default=8)
    ap.add_argument("--probe-timeout", type=int, default=240)
    ap.add_argument("--probe-fib19-only", action="store_true", help="shard probe: only the bench workload, not the 2^24-row and 2^26-row Poseidon252 traces (BASELINE configs 3-5)")
    ap.add_argument("--no-local-probe", action="store_true", help="shard probe: skip the in-process variant (rank 0's child driving all N GPUs from N host threads)")
    ap.add_argument("--probe-local", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--shard-probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--probe-out", help=argparse.SUPPRESS)
    ap.add_argument("--probe-id-file", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.shard_probe:
        return shard_probe(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # N > 1, replicas mode: besides the contract's weak-scaling number, measure ONE proof over all N GPUs (shard group, strong scaling) in
    # child processes first — bounded, killed on timeout, never allowed to cost the main line.
    shard_probe_result = None
    if world > 1 and not args.shard and not args.no_shard_probe and world & (world - 1) == 0:
        try:
            shard_probe_result = run_shard_probe(args, rank, world)
        except Exception as e:
            shard_probe_result = {"n_gpus": world, "error": repr(e)}

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
        # control plane only: rank 0's RCCL unique id reaches the others through torch.distributed; every data-path exchange of the proof
        # is issued by libbfhip itself on the context's stream (RCCL over xGMI, device buffers on both ends)
        dev = torch.device("cuda", device) if args.dist_backend == "nccl" else None
        ctx.join_rccl_group(replicas.share_unique_id(dist, pkg.rccl_unique_id, dev), rank, world)

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
    pin = lambda: pinned_host_thread(args.pin and args.inflight == 1, device, local_rank, world)      # noqa: E731
    with pin():
        dt, (proof, phases) = replicas.timed_region(one_step, args.steps, args.warmup, dist=dist, sync_fn=sync,
                                                    backend_tensor=cuda_t, on_timed_start=start_events)
    # replicas: every rank proves its own trace (units add up); shard group: all ranks prove the same one
    total_cells = trace.cells if sharded else replicas.aggregate_units(trace.cells * args.inflight, dist=dist, backend_tensor=cuda_t)

    # ---- parity: the proof timed last against the committed digest of the CPU oracle's proof of this workload (same conventions) ------
    digest = hashlib.sha256(proof).hexdigest()
    want = next((d for d in committed_digests().values() if tuple(d.get("conventions", ())) == conv and d.get("log_max_rows") == args.log_max_rows), None)
    parity_checked = bool(want is not None and want["sha256"] == digest and want["proof_bytes"] == len(proof))
    verified, why = pkg.verify_brainfuck(proof, args.log_max_rows)

    roofline, fft = None, None
    if not args.no_kernel_events:
        rep = profile_report(lib, ctx)
        lib.bfhip_profile_enable(ctx._h, 0)
        name, d = max(rep.items(), key=lambda kv: kv[1]["total_ms"])
        avg_ms = d["total_ms"] / d["calls"]
        gbs = d["bytes"] / d["calls"] / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        import glob
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))   # from the separate rocprofv3 --pmc passes of the latest round
        if pmc_files:
            # The counter passes are a separate command (rocprofv3 --pmc serialises the dispatches: it cannot run inside a timed region), so the
            # figure is read from the latest committed file — and only trusted while the kernel it was taken on is the kernel that ran here:
            # tools/pmc_traffic.py records the SHA-256 of the kernel sources; a mismatch (or a file without the record) reports null.
            pmc = json.load(open(pmc_files[-1]))
            want_src = pmc.get("_kernel_sources_sha256")
            have_src = kernel_sources_sha256()
            if want_src == have_src:
                traffic = pmc.get(name, {}).get("hbm_bytes_per_launch")
                traffic_src = os.path.relpath(pmc_files[-1], ROOT) + " (separate rocprofv3 --pmc passes of the same command on the same kernel sources; not collected in this run)"
            else:
                traffic_src = os.path.relpath(pmc_files[-1], ROOT) + " is STALE: taken on other kernel sources (csrc/merkle.hip, csrc/kernels.h changed since) — traffic not reported; rerun tools/profile_round.sh pmc"
        hbm = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
               "algorithmic_bytes_per_launch": round(d["bytes"] / d["calls"])}
        common = {"kernel": name, "traffic": traffic, "traffic_source": traffic_src, "launches": d["calls"], "avg_launch_us": round(avg_ms * 1e3, 2),
                  "kernels_ms_per_step": {k: round(v["total_ms"] / args.steps, 3) for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"])}}
        if name.startswith("k_merkle_layer") and d.get("units", 0) > 0:
            # The Merkle kernel is integer-VALU bound: ~977 lane-ops per Blake2s compression, one compression per
            # 64 message bytes. Compressions are counted from the launch shapes of this very run (prof.hip `units`), not a constant.
            comp_per_step = d["units"] / args.steps
            tops = d["units"] * VALU_OPS_PER_COMPRESSION / (d["total_ms"] * 1e-3) / 1e12
            roofline = {**common, "bound": "valu", "achieved": round(tops, 2), "peak": round(VALU_PEAK_TOPS, 2), "unit": "Tops/s (int32 VALU lane-ops)",
                        "frac": round(tops / VALU_PEAK_TOPS, 4), "compressions_per_proof": round(comp_per_step),
                        "G_compressions_per_s": round(d["units"] / (d["total_ms"] * 1e-3) / 1e9, 2), "valu_ops_per_compression": VALU_OPS_PER_COMPRESSION,
                        "hbm": hbm}
        else:
            roofline = {**common, **hbm}
        # ---- the circle-FFT kernels (north-star: >= 60 % HBM on the FFT kernel): one extra UNTIMED proof with every kernel bracketed ----
        if world == 1:
            lib.bfhip_profile_enable(ctx._h, 1)
            lib.bfhip_profile_reset(ctx._h)
            trace.prove(args.log_max_rows)
            full = profile_report(lib, ctx)
            lib.bfhip_profile_enable(ctx._h, 0)
            fk = {k: v for k, v in full.items() if k.startswith("k_fft")}
            if fk:
                tot_ms = sum(v["total_ms"] for v in fk.values())
                fft = {"kernels": {k: {"ms_per_proof": round(v["total_ms"], 3), "launches": v["calls"], "moved_GBps": round(v["bytes"] / v["total_ms"] / 1e6, 1),
                                       "moved_frac_of_hbm_peak": round(v["bytes"] / v["total_ms"] / 1e6 / HBM_PEAK_GBS, 4)} for k, v in sorted(fk.items())},
                       "ms_per_proof": round(tot_ms, 3),
                       "moved_GBps": round(sum(v["bytes"] for v in fk.values()) / tot_ms / 1e6, 1),
                       "algorithmic_GBps": round(sum(v["units"] for v in fk.values()) / tot_ms / 1e6, 1),
                       "algorithmic_frac_of_hbm_peak": round(sum(v["units"] for v in fk.values()) / tot_ms / 1e6 / HBM_PEAK_GBS, 4),
                       "note": "in-proof mix of column sizes (most launches are small); algorithmic bytes = 8N per interpolated, 12N per extended column (SURVEY.md section 8(d)); "
                               "the 128 x 2^24 kernel run is tools/fft_roofline.py -> profiles/"}
                roofline["all_kernels_ms_per_proof_instrumented"] = {k: round(v["total_ms"], 3) for k, v in sorted(full.items(), key=lambda kv: -kv[1]["total_ms"])}
                # every Blake2s compression of the proof's trees, whichever kernel ran it (fixed by the protocol): k_merkle_layer + the
                # multi-level kernels of the small end (k_merkle_subtree, k_merkle_top, k_fri_layer, k_fri_tail)
                roofline["compressions_per_proof_all_merkle_kernels"] = round(sum(v.get("units", 0) for k, v in full.items() if k in ("k_merkle_layer", "k_merkle_subtree", "k_merkle_top", "k_fri_tail", "k_fri_layer")))

    # ---- N = 1: throughput with two proofs in flight (one context + host thread each): the VALU-bound hashing of one proof overlaps the
    # HBM-bound transforms of the other. Reported beside `value`, never as it.
    pipelined = None
    if world == 1 and args.inflight == 1 and not args.no_sweep:
        try:
            import threading
            c2 = pkg.Context(device, max_log_domain=args.log_max_rows + 2)
            t2 = pkg.Trace(c2, FIB19, b"")
            pair = [(ctx, trace), (c2, t2)]
            def run_pair(k):
                th = [threading.Thread(target=lambda tr=tr: [tr.prove(args.log_max_rows, want_json=False) for _ in range(k)]) for _, tr in pair]
                [t.start() for t in th]; [t.join() for t in th]
                ctx.sync(); c2.sync()
            run_pair(1)
            t0 = time.perf_counter(); run_pair(5); dtp = time.perf_counter() - t0
            pipelined = {"proofs_in_flight": 2, "ms_per_proof": round(dtp / 10 * 1e3, 3), "cells_per_s": trace.cells * 10 / dtp}
            t2.close(); c2.close()
        except Exception as e:
            pipelined = {"error": repr(e)}

    sweep = None
    if world == 1 and not args.no_sweep and rank == 0:
        trace_cells = trace.cells
        try:
            with pin():
                sweep = run_sweep(pkg, device, args.sweep_steps, [int(v) for v in args.sweep_logs.split(",")])
        except Exception as e:      # the sweep must never cost the headline line
            sweep = {"error": repr(e)}
    poseidon = None
    if world == 1 and not args.no_sweep and not args.no_poseidon and rank == 0:
        try:
            with pin():
                poseidon = run_poseidon_point(pkg, device, args.poseidon_log)
        except Exception as e:
            poseidon = {"error": repr(e)}

    cells = trace.cells
    if rank == 0:
        headline22 = next((p for p in sweep if p["log_domain_rows"] == 22), None) if isinstance(sweep, list) else None
        out = {
            "metric": "trace cells committed+proved/sec",
            "value": total_cells * args.steps / dt,
            "unit": "trace cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            # BASELINE.json's metric is quoted "at 2^22 rows": that point of the sweep, promoted (value above is the larger fib19 workload)
            "metric_point": ({"workload": "synthetic nested-counter trace, Memory component 2^22 domain rows, LOG_MAX_ROWS 22 (BASELINE metric 'at 2^22 rows')",
                              "value": headline22["cells_per_s"], "unit": "trace cells/s", "ms_per_proof": headline22["ms_per_proof"], "cells": headline22["cells"],
                              "proof_sha256": headline22["proof_sha256"], "verified": headline22["verified"]} if headline22 else None),
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "u32 (M31 / QM31 modular arithmetic)",
            "data": "fib19.bf execution trace (199246 VM steps), synthetic in the sense of the contract: a bundled program, no external data",
            "parity_checked": parity_checked,
            "parity": {"proof_sha256": digest, "proof_bytes": len(proof), "expected_sha256": want["sha256"] if want else None,
                       "expected_from": "tests/golden/fib19_lmr24_oracle_proof.json (CPU oracle's proof of this workload under the same conventions)" if want else None,
                       "conventions": list(conv), "own_verifier_accepts": bool(verified)},
            "config": {"workload": "fib19.bf, largest component 2^20 table rows = 2^24 domain rows, Blake2s Merkle, 1 proof per step",
                       "log_max_rows": args.log_max_rows, "cells_per_proof": cells, "main_cells": trace.main_cells, "interaction_cells": trace.interaction_cells,
                       "component_log_sizes": trace.log_sizes, "parallelism": ("shard group: one proof over all ranks (column-sharded transforms, row-sharded Merkle/constraints/quotients/folds, RCCL)" if sharded else "replicas") if world > 1 else "single", "proofs_in_flight_per_gpu": args.inflight, "host_thread_pinned_to_cpu": pinned_host_thread.cpu, "preprocessed_tree": "reused across proofs" if args.reuse_preprocessed else "recommitted every proof (as the reference)",
                       "proof_bytes": len(proof), "phase_ms_last_step": {k: round(v * 1e3, 2) for k, v in phases.items()},
                       "headline_2^22": ({"cells_per_s": headline22["cells_per_s"], "ms_per_proof": headline22["ms_per_proof"], "cells": headline22["cells"],
                                          "workload": "synthetic nested-counter trace, Memory component 2^22 domain rows, LOG_MAX_ROWS 22 (BASELINE metric 'at 2^22 rows')",
                                          **{k2: headline22[k2] for k2 in ("roofline", "gpu_busy_frac_estimate", "sum_of_kernel_ms", "kernels_ms_per_proof_instrumented") if k2 in headline22}}
                                         if headline22 else None)},
            "roofline": roofline,
            "fft": fft,
            "pipelined": pipelined,
            "sweep": sweep,
            "poseidon252": poseidon,
        }
        if shard_probe_result is not None:
            # strong scaling beside the weak-scaling value: the same workload proved ONCE by all N GPUs together
            out["shard_group"] = shard_probe_result
            out["strong_scaling"] = strong_scaling_summary(shard_probe_result, world)
        if world == 1 and not args.no_cpu_baseline:
            full = args.cpu_baseline == "full"
            if args.cpu_baseline == "auto":
                avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
                try:
                    free_gb = int(next(l for l in open("/proc/meminfo") if l.startswith("MemAvailable")).split()[1]) / 1e6
                except Exception:
                    free_gb = 0.0
                full = avail >= 32 and free_gb >= 48
            out["cpu_baseline"] = cpu_baseline(cells, full=full)
            # a reported ratio, not a quality claim: the port is scalar C++ with OpenMP loops, the Rust SIMD prover cannot be built here
            out["cpu_baseline"]["gpu_over_port"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
            if full and "proof_sha256" in out["cpu_baseline"]:
                out["cpu_baseline"]["proof_identical_to_gpu"] = out["cpu_baseline"]["proof_sha256"] == digest
            try:
                sb = simd_bound(dt / args.steps, trace.log_sizes, args.log_max_rows)
                if "seconds_lower_bound" in sb:
                    sb["cells_per_s_upper_bound"] = cells / sb["seconds_lower_bound"]["total"]
                out["cpu_baseline"]["simd_bound"] = sb
            except Exception as e:
                out["cpu_baseline"]["simd_bound"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if args.reuse_preprocessed:
        lib.bfhip_ctx_reuse_preprocessed(ctx._h, 0)
    if sharded:
        ctx.leave_group()
    for c2, t2 in extra:
        t2.close(); c2.close()
    trace.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
