#!/usr/bin/env python3
"""One workload proved by N ranks of an in-process shard group on ONE GPU — the program that runs under `rocprofv3 --kernel-trace` to find
out WHICH kernels every rank of a group repeats (tools/shard_redundancy.py reads the traces). Every rank is one host thread, so the
trace's Thread_Id column attributes each launch to its rank; the mapping is printed.

  python3 tools/shard_kernels.py N [fib19 | LOG] [--steps K] [--poseidon] [--overlap MASK]

Prints one JSON line: ms per proof (wall, ranks time-sharing the GPU), number of proofs every rank ran (all of them are in the trace),
native thread id of every rank, SHA-256 of the proof, per-phase host times of rank 0."""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", type=int)
    ap.add_argument("what", nargs="?", default="fib19")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--poseidon", action="store_true")
    ap.add_argument("--units", action="store_true", help="count the hash units (Blake2s compressions / Hades permutations) every rank computes: the library's own "
                    "per-kernel accounting (bfhip_profile_enable mode 1) over the timed proofs; slows the proofs down")
    ap.add_argument("--overlap", type=int, default=-1, help="bfhip_ctx_set_overlap mask of every rank (-1: the library's default)")
    args = ap.parse_args()
    pkg = bench.load_package()
    if args.what == "fib19":
        code, lmr, name = bench.FIB19, 24, "fib19.bf"
    else:
        k = int(args.what)
        code, lmr, name = bench.sweep_program(k), k, f"synthetic 2^{k} domain rows"
    conv = (0, 0, 0, 1 if args.poseidon else 0)
    n = args.n
    group = pkg.LocalGroup(n) if n > 1 else None
    ctxs = [pkg.Context(0, max_log_domain=lmr + 2) for _ in range(n)]
    for c in ctxs:
        c.set_conventions(*conv)
    traces = [pkg.Trace(c, code, b"") for c in ctxs]
    proofs, phases, times, tids, units = [None] * n, [None] * n, [0.0] * n, [0] * n, [None] * n
    gate = threading.Barrier(n)

    def work(r):
        tids[r] = threading.get_native_id()
        if group:
            ctxs[r].join_local_group(group, r)
            if args.overlap >= 0:
                ctxs[r].set_overlap(args.overlap)
        for _ in range(args.warmup):
            traces[r].prove(lmr)
        ctxs[r].sync()
        if args.units:
            pkg.lib().bfhip_profile_enable(ctxs[r]._h, 1); pkg.lib().bfhip_profile_reset(ctxs[r]._h)
        gate.wait()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            proofs[r], phases[r] = traces[r].prove(lmr)
        ctxs[r].sync()
        times[r] = (time.perf_counter() - t0) / args.steps
        if args.units:
            rep = bench.profile_report(pkg.lib(), ctxs[r])
            pkg.lib().bfhip_profile_enable(ctxs[r]._h, 0)
            units[r] = {k: round(v["units"] / args.steps) for k, v in rep.items() if v.get("units", 0) > 0 and k.startswith(("k_merkle", "k_fri"))}

    th = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    [t.start() for t in th]; [t.join() for t in th]
    stats = ctxs[0].group_stats() if group else None
    mem = [c.memory() for c in ctxs]
    for r in range(n):
        if group:
            ctxs[r].leave_group()
        traces[r].close(); ctxs[r].close()
    if group:
        group.close()
    assert all(p == proofs[0] for p in proofs)
    print(json.dumps({"workload": name + (", Poseidon252" if args.poseidon else ", Blake2s"), "ranks_on_one_gpu": n, "ms_per_proof_wall": round(max(times) * 1e3, 3),
                      "proofs_in_trace_per_rank": args.warmup + args.steps, "rank_thread_ids": tids, "proof_sha256": hashlib.sha256(proofs[0]).hexdigest(),
                      "rank0_phase_ms_last_proof": {k: round(v * 1e3, 2) for k, v in phases[0].items()},
                      "rank0_group_stats_total": stats,
                      **({"hash_units_per_proof_by_rank": units, "hash_units_per_proof_all_ranks": sum(sum(u.values()) for u in units)} if args.units else {}),
                      "arena_peak_GB_per_rank": [round(m["arena_peak"] / 2**30, 2) for m in mem],
                      "device_GB_reserved_all_ranks": round(sum(m["arena_reserved"] + m["twiddles"] for m in mem) / 2**30, 1)}))


if __name__ == "__main__":
    main()
