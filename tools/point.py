#!/usr/bin/env python3
"""One workload, proved `steps` times on cuda:0 — the unit the per-size work of a round is measured on (and what runs under rocprofv3 for
the per-size kernel traces in profiles/).

  python3 tools/point.py 22            synthetic nested-counter trace, Memory component 2^22 domain rows, LOG_MAX_ROWS 22 (bench.py sweep point)
  python3 tools/point.py fib19         the bench workload (fib19.bf, LOG_MAX_ROWS 24)
  options: --steps N (default 10) --warmup W (default 2) --conventions a,b,c,d

Prints one JSON line: ms per proof (mean and min over steps), phase split of the last proof, SHA-256 of the proof, verifier verdict."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--conventions", default="0,0,0,0")
    ap.add_argument("--times", action="store_true", help="also print the sorted per-proof times (host jitter shows as a tail)")
    args = ap.parse_args()
    pkg = bench.load_package()
    conv = tuple(int(v) for v in args.conventions.split(","))
    if args.what == "fib19":
        code, lmr, name = bench.FIB19, 24, "fib19.bf"
    else:
        k = int(args.what)
        code, lmr, name = bench.sweep_program(k), k, f"synthetic 2^{k} domain rows"
    c = pkg.Context(0, max_log_domain=lmr + 2)
    try:
        c.set_conventions(*conv)
        tr = pkg.Trace(c, code, b"")
        try:
            for _ in range(args.warmup):
                tr.prove(lmr)
            c.sync()
            # a proof is complete when prove() returns (its last bytes were read from the GPU): no stream synchronisation per proof, as in
            # bench.py's timed loops; ms_per_proof = wall time of the loop / steps, ms_min = the fastest single call
            times = []
            t_loop = time.perf_counter()
            for _ in range(args.steps):
                t0 = time.perf_counter()
                proof, phases = tr.prove(lmr)
                times.append(time.perf_counter() - t0)
            c.sync()
            t_loop = time.perf_counter() - t_loop
            ok, why = pkg.verify_brainfuck(proof, lmr, conv)
            print(json.dumps({"workload": name, "log_max_rows": lmr, "cells": tr.cells, "steps": args.steps,
                              "ms_per_proof": round(1e3 * t_loop / args.steps, 3), "ms_min": round(1e3 * min(times), 3),
                              "cells_per_s": tr.cells / (t_loop / args.steps),
                              "phase_ms": {k: round(v * 1e3, 3) for k, v in phases.items()},
                              "proof_bytes": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(), "verified": bool(ok), "why": why,
                              **({"ms_sorted": [round(1e3 * t, 3) for t in sorted(times)]} if args.times else {})}))
        finally:
            tr.close()
    finally:
        c.close()


if __name__ == "__main__":
    main()
