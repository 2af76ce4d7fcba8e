#!/usr/bin/env python3
"""Proofs in flight behind ONE caller thread: bfhip_prove_batch over a pool of k sub-contexts (include/bfhip.h bfhip_pool_*), the same resident
trace n times per batch.

  python3 tools/pool_rate.py 22 --in-flight 1,2,3 --preprocessed 0,1,2 --batch 12 --batches 4
  python3 tools/pool_rate.py fib19 --in-flight 1,2 --batch 6

Every (k, mode) configuration runs in a child process of its own, started before this process touches the GPU (the hardware queues a process's
streams get depend on its history). One JSON line per configuration: ms per proof = wall time of the timed batches / proofs, cells/s, SHA-256 of every
proof (all equal), and `plain` = the same trace proved one at a time on a plain context in that child (the single-proof reference of the same box)."""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(what, k, mode, batch, batches, warm):
    import bench
    pkg = bench.load_package()
    if what == "fib19":
        code, lmr, name = bench.FIB19, 24, "fib19.bf"
    else:
        kk = int(what)
        code, lmr, name = bench.sweep_program(kk), kk, f"synthetic 2^{kk} domain rows"
    out = {"workload": name, "log_max_rows": lmr, "in_flight": k, "preprocessed_mode": mode, "batch": batch, "batches": batches}
    # the single-proof reference first, on a plain context that is closed before the pool exists
    c = pkg.Context(0, max_log_domain=lmr + 2)
    tr = pkg.Trace(c, code, b"")
    for _ in range(2):
        tr.prove(lmr)
    n1 = max(4, batch)
    t0 = time.perf_counter()
    for _ in range(n1):
        proof, _ = tr.prove(lmr)
    c.sync()
    out["plain_ms_per_proof"] = round(1e3 * (time.perf_counter() - t0) / n1, 3)
    sha1 = hashlib.sha256(proof).hexdigest()
    cells = tr.cells
    tr.close(); c.close()
    pool = pkg.Pool(0, n_in_flight=k, max_log_domain=lmr + 2, preprocessed=mode)
    try:
        tr = pkg.Trace(pool.ctx(0), code, b"")
        traces = [tr] * batch
        for _ in range(warm):
            pool.prove_batch(traces, lmr)
        secs = []
        t0 = time.perf_counter()
        for _ in range(batches):
            proofs, info = pool.prove_batch(traces, lmr)
            secs.append(info["batch_seconds"])
        dt = time.perf_counter() - t0
        shas = {hashlib.sha256(p).hexdigest() for p in proofs}
        if os.environ.get("POOL_RATE_PROGRAMS") == "1":
            # host-inclusive: bfhip_prove_batch_brainfuck — VM run, table build and upload of every proof inside its worker, beside the other workers' GPU work
            progs = [(code, b"")] * batch
            pool.prove_batch_brainfuck(progs, lmr)
            t1 = time.perf_counter()
            for _ in range(batches):
                pp, _ = pool.prove_batch_brainfuck(progs, lmr)
            dtp = time.perf_counter() - t1
            out["from_program_text"] = {"ms_per_proof": round(1e3 * dtp / (batch * batches), 3), "cells_per_s": cells * batch * batches / dtp,
                                        "identical_to_plain": {hashlib.sha256(p).hexdigest() for p in pp} == {sha1},
                                        "what": "bfhip_prove_batch_brainfuck: VM + 13 table builders + upload + proof per program, end to end (PCIe inclusive)"}
        out.update(ms_per_proof=round(1e3 * dt / (batch * batches), 3), cells_per_s=cells * batch * batches / dt, cells=cells,
                   batch_ms=[round(1e3 * s, 2) for s in secs], proof_ms_in_batch=round(1e3 * sum(info["seconds"]) / batch, 3),
                   identical_to_plain=shas == {sha1}, proof_sha256=sha1, gain_vs_plain=round(out["plain_ms_per_proof"] / (1e3 * dt / (batch * batches)), 3))
        tr.close()
    finally:
        pool.close()
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what")
    ap.add_argument("--in-flight", default="1,2,3")
    ap.add_argument("--preprocessed", default="1")
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--warm", type=int, default=1)
    ap.add_argument("--child", default=None)
    args = ap.parse_args()
    if args.child:
        k, mode = (int(v) for v in args.child.split(","))
        return child(args.what, k, mode, args.batch, args.batches, args.warm)
    for k in (int(v) for v in args.in_flight.split(",")):
        for mode in (int(v) for v in args.preprocessed.split(",")):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), args.what, "--child", f"{k},{mode}", "--batch", str(args.batch), "--batches", str(args.batches),
                                "--warm", str(args.warm)], capture_output=True, text=True, timeout=900)
            line = r.stdout.strip().split("\n")[-1] if r.stdout.strip() else ""
            print(line if r.returncode == 0 and line else json.dumps({"in_flight": k, "preprocessed_mode": mode, "error": (r.stderr or r.stdout)[-400:]}), flush=True)


if __name__ == "__main__":
    main()
