"""Measurements that ride in the bench line beside the contract path: the sweep, the Poseidon252 point, proofs in flight (see tools/benchlib/__init__.py)."""
import hashlib
import json
import os
import subprocess
import sys
import time

from .roofline import point_roofline
from .workloads import BENCH, FIB19, sweep_program


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


# ---- proofs in flight (N = 1): through the library's pool, ONE caller thread (include/bfhip.h bfhip_pool_* / bfhip_prove_batch) -----------------------
# (name, sweep log or None = the bench workload, LOG_MAX_ROWS or None = --log-max-rows, proofs per batch)
PIPELINED_WORK = [("fib19", None, None, 6), ("2^22_rows", 22, 22, 12), ("2^20_rows", 20, 20, 24)]


def run_pipelined_one(pkg, device, code, lmr, k, mode, batch):
    """One configuration, in a process of its own. k = 1: one proof at a time on a plain context (one call = one proof, mod.rs:471-735: the reference
    of the gain). k > 1: bfhip_prove_batch over a pool of k sub-contexts — one caller thread, the library's own workers; mode = how the pool treats the
    preprocessed tree (1 = one commitment per batch, the pool's default; 0 = every proof recommits it like the reference). Whole batches are timed until
    0.5 s have passed (a 2^20-row configuration is over in 50 ms otherwise, before the clocks have settled)."""
    if k == 1:
        c = pkg.Context(device, max_log_domain=lmr + 2)
        tr = pkg.Trace(c, code, b"")
        try:
            for _ in range(3):
                tr.prove(lmr, want_json=False)
            n, t0 = 0, time.perf_counter()
            while n == 0 or time.perf_counter() - t0 < 0.5:
                for _ in range(batch):
                    tr.prove(lmr, want_json=False)
                n += batch
            c.sync()
            dt = time.perf_counter() - t0
            proof, _ = tr.prove(lmr)
            shas, cells = [hashlib.sha256(proof).hexdigest()], tr.cells
        finally:
            tr.close(); c.close()
        what = "one proof at a time on a plain context (bfhip_prove_trace)"
    else:
        pool = pkg.Pool(device, n_in_flight=k, max_log_domain=lmr + 2, preprocessed=mode)
        tr = pkg.Trace(pool.ctx(0), code, b"")
        try:
            traces = [tr] * batch
            pool.prove_batch(traces, lmr, want_json=False)          # warm-up (arena growth, first-proof setup, clocks)
            n, t0 = 0, time.perf_counter()
            while n == 0 or time.perf_counter() - t0 < 0.5:
                pool.prove_batch(traces, lmr, want_json=False)
                n += batch
            dt = time.perf_counter() - t0
            proofs, _ = pool.prove_batch(traces[:k], lmr)           # the bytes: one more proof per worker with the JSON kept
            shas, cells = [hashlib.sha256(p).hexdigest() for p in proofs], tr.cells
            shared = any(pool.ctx(i).last_proof_flags()["shared_preprocessed"] for i in range(k))
        finally:
            tr.close(); pool.close()
        what = (f"bfhip_prove_batch: batches of {batch} proofs over a pool of {k} sub-contexts, one caller thread; preprocessed tree "
                + ("committed once per batch" if mode == 1 else "recommitted by every proof (as the reference)") + (", shared tree in use" if shared else ""))
    ms = dt / n * 1e3
    return {"ms_per_proof": round(ms, 3), "cells_per_s": cells / (ms * 1e-3), "proofs_timed": n, "proof_sha256": shas, "all_same_proof": len(set(shas)) == 1, "how": what}


def run_pipelined(args):
    """{fib19, 2^22 rows, 2^20 rows} x {1, 2, 3 proofs in flight}, every configuration in a CHILD PROCESS of its own, started before this process
    touches the GPU (the hardware queues a process's streams get depend on its history: profiles/r05_inflight_history.txt). Since round 6 the proofs
    in flight are the LIBRARY's (a pool behind one caller thread), not k Python threads over k contexts. At the metric's size (2^22 rows) the
    pool is also measured with every proof recommitting the preprocessed tree (`recommitted`), which is what the reference's prove_brainfuck does."""
    out = {"what": "k proofs in flight per GPU = bfhip_prove_batch over a pool of k sub-contexts (one caller thread), in a fresh process; in_flight_1 = one proof at a time on a plain context; "
                   "ms_per_proof = wall time / proofs completed"}

    def child(name, k, mode):
        cmd = [sys.executable, BENCH, "--pipelined-child", f"{name}:{k}:{mode}", "--log-max-rows", str(args.log_max_rows)] + (["--device", str(args.device)] if args.device is not None else [])
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{")), None)
            return json.loads(line) if line else {"error": f"child exited {r.returncode}: {r.stderr[-300:]}"}
        except Exception as e:
            return {"error": repr(e)}

    for name, _, _, _ in PIPELINED_WORK:
        row = {f"in_flight_{k}": child(name, k, 1) for k in (1, 2, 3)}
        if name == "2^22_rows":
            for k in (2, 3):
                row[f"in_flight_{k}"]["recommitted"] = {kk: vv for kk, vv in child(name, k, 0).items() if kk in ("ms_per_proof", "cells_per_s", "proofs_timed", "error", "how")}
        base = row["in_flight_1"]
        for k in (2, 3):
            cur = row[f"in_flight_{k}"]
            if "ms_per_proof" in base and "ms_per_proof" in cur:
                cur["gain_vs_1"] = round(base["ms_per_proof"] / cur["ms_per_proof"], 3)
                cur["same_proof_as_1"] = cur["proof_sha256"][0] == base["proof_sha256"][0]
                if "ms_per_proof" in cur.get("recommitted", {}):
                    cur["recommitted"]["gain_vs_1"] = round(base["ms_per_proof"] / cur["recommitted"]["ms_per_proof"], 3)
        out[name] = row
    return out


def pipelined_child_main(args):
    """bench.py --pipelined-child name:k:mode — ONE proofs-in-flight configuration in a process of its own (run_pipelined)."""
    from .workloads import load_package, pick_device
    name, k, mode = args.pipelined_child.split(":")
    _, sweep_log, lmr, batch = next(w for w in PIPELINED_WORK if w[0] == name)
    pkg = load_package()
    code, lmr = (FIB19, args.log_max_rows) if sweep_log is None else (sweep_program(sweep_log), lmr)
    print(json.dumps(run_pipelined_one(pkg, pick_device(0, pkg.device_count(), args.device), code, lmr, int(k), int(mode), batch)), flush=True)
    return 0


def batch_summary(pipelined):
    """config.batch of the line: the metric's own size (2^22 rows) through the pool, from one caller thread — inside `config` because the driver's record
    keeps `config` whole (BENCH_r05.json lost the top-level `pipelined` key)."""
    row = pipelined.get("2^22_rows") if isinstance(pipelined, dict) else None
    if not row:
        return None
    pick = lambda d: {k: d[k] for k in ("ms_per_proof", "cells_per_s", "gain_vs_1", "same_proof_as_1", "error") if k in d} if isinstance(d, dict) else None      # noqa: E731
    out = {"what": "bfhip_prove_batch at the metric's size (synthetic trace, 2^22 domain rows, LOG_MAX_ROWS 22): batches of 12 proofs through a pool of k sub-contexts, ONE caller thread; "
                   "shared = one preprocessed commitment per batch (pool default), recommitted = every proof commits its own (as the reference); value itself stays single-proof",
           "one_at_a_time": pick(row.get("in_flight_1"))}
    for k in (2, 3):
        cur = row.get(f"in_flight_{k}") or {}
        out[f"in_flight_{k}"] = {"shared_preprocessed": pick(cur), "recommitted_preprocessed": pick(cur.get("recommitted"))}
    best = [v for k in (2, 3) for v in [(row.get(f"in_flight_{k}") or {}).get("ms_per_proof")] if v]
    if best:
        out["ms_per_proof"] = min(best)
        out["cells_per_s"] = max((row[f"in_flight_{k}"]["cells_per_s"] for k in (2, 3) if "cells_per_s" in (row.get(f"in_flight_{k}") or {})), default=None)
    return out
