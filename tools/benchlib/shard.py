"""One proof over N GPUs in CHILD processes (bench.py --shard-probe): the in-process transport probe and, on request, per-rank RCCL children."""
import hashlib
import json
import os
import sys
import time

from .workloads import BENCH, FIB19, committed_digests, load_package, pick_device, sweep_program


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
                if overlap is not None:                  # None: the library's default (exchange on the partner stream when the group spans GPUs)
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
        cmd = [sys.executable, BENCH, "--shard-probe", "--probe-out", out_path, "--probe-id-file", f"{base}.id", "--probe-steps", str(args.probe_steps),
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
    # round 5: the RCCL group's proofs are the headline of the main processes themselves; the per-rank RCCL children run on request only
    result = run_child([], f"{base}_rank{rank}.json") if args.rccl_child_probe else {"n_gpus": world, "stages": {}}
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
