#!/usr/bin/env python3
"""k proofs in flight on one GPU — k FRESH contexts (stream, arena, staging ring each) and one host thread each, proving the same workload
back to back: the program to put under `rocprofv3 --kernel-trace` for profiles/r05_2p22_inflight2_timeline_gaps.txt
(tools/timeline_gaps.py --tail-ms reads the trace), and to time un-profiled beside it. Prints one JSON line.

  python3 tools/inflight_profile.py 22 2 [--rounds 6] [--warmup 2]        workload: 20..26 (synthetic 2^k-row trace) or fib19; k in flight"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("k", type=int)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--min-seconds", type=float, default=0.5, help="repeat the timed wave until this much time has been measured (0: exactly one wave, for traces)")
    args = ap.parse_args()
    import bench
    pkg = bench.load_package()
    if args.workload == "fib19":
        code, lmr = bench.FIB19, 24
    else:
        lmr = int(args.workload)
        code = bench.sweep_program(lmr)
    ctxs = [pkg.Context(0, max_log_domain=lmr + 2) for _ in range(args.k)]
    traces = [pkg.Trace(c, code, b"") for c in ctxs]
    shas = [None] * args.k

    def wave(n, keep):
        def run(i):
            for _ in range(n):
                proof, _ = traces[i].prove(lmr, want_json=keep)
            if keep:
                shas[i] = hashlib.sha256(proof).hexdigest()
        th = [threading.Thread(target=run, args=(i,)) for i in range(args.k)]
        [t.start() for t in th]; [t.join() for t in th]
        for c in ctxs:
            c.sync()

    wave(args.warmup, False)
    waves, t0 = 0, time.perf_counter()
    while waves == 0 or time.perf_counter() - t0 < args.min_seconds:      # short configurations are repeated until the window is long enough for settled clocks
        wave(args.rounds, False); waves += 1
    dt = time.perf_counter() - t0
    wave(1, True)
    proofs = waves * args.rounds * args.k
    print(json.dumps({"workload": args.workload, "log_max_rows": lmr, "in_flight": args.k, "rounds": args.rounds, "proofs_timed": proofs, "ms_per_proof": round(dt / proofs * 1e3, 3),
                      "timed_window_ms": round(dt * 1e3, 2), "cells_per_s": traces[0].cells * proofs / dt, "proof_sha256": shas}))
    for t in traces:
        t.close()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
