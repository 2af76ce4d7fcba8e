"""The N > 1 headline of bench.py: ONE proof over the N ranks' GPUs (shard group over RCCL, strong scaling), with the N independent replicas measured
first — they are the `replicas` field, and the line this run prints if the group cannot be formed, fails, or never comes back."""
import hashlib
import json
import os
import sys
import threading
import time

from .shard import probe_run_stages
from .workloads import FIB19, sweep_program, want_digest


class Ranks:
    """This process's place among the ranks and the timing channel (torch.distributed: barrier, max-over-ranks, the 128-byte unique id)."""

    def __init__(self, args, rank, local_rank, world, torch=None, dist=None):
        self.args, self.rank, self.local_rank, self.world, self.torch, self.dist = args, rank, local_rank, world, torch, dist
        self.t_start = time.time()
        self.cuda_t = (lambda v: torch.tensor([v], dtype=torch.float64, device="cuda")) if (dist is not None and args.dist_backend == "nccl") else None

    def note(self, msg):
        """N > 1: one stderr line per stage and rank — where a multi-GPU run is when something hangs (the JSON line stays the only stdout)."""
        if self.world > 1:
            print(f"bench.py[rank {self.rank}/{self.world} +{time.time() - self.t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    def over_ranks(self, value, op):
        """max / min / sum of a number over the ranks."""
        if self.dist is None:
            return value
        t = self.torch.tensor([float(value)], dtype=self.torch.float64) if self.cuda_t is None else self.cuda_t(float(value))
        self.dist.all_reduce(t, op={"max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN, "sum": self.dist.ReduceOp.SUM}[op])
        return float(t.item())

    def agree(self, ok):
        """True only if the step succeeded on EVERY rank (the ranks must take the same path afterwards)."""
        return self.over_ranks(1.0 if ok else 0.0, "min") > 0.5

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()


def replicas_fallback_line(args, R, trace, conv, replica_line, reason):
    """The line of a multi-GPU run whose shard group is unusable: the replicas (weak scaling), measured under the contract's protocol before the group formed,
    flagged with shard_group_error."""
    want = want_digest(conv, args.log_max_rows)
    return {"metric": "trace cells committed+proved/sec", "value": replica_line["value"], "unit": "trace cells/s", "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": replica_line["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 (M31 / QM31 modular arithmetic)",
            "data": "fib19.bf execution trace (199246 VM steps), synthetic in the sense of the contract: a bundled program, no external data",
            "parity_checked": bool(want is not None and want["sha256"] == replica_line["proof_sha256"]),
            "config": {"workload": "fib19.bf (BASELINE config 2; 2^24 domain rows, Blake2s Merkle), 1 proof per step and GPU", "log_max_rows": args.log_max_rows,
                       "cells_per_proof": trace.cells, "parallelism": "replicas"},
            "roofline": None, "replicas": replica_line, "shard_group_error": reason}


def run(args, R, pkg, replicas, ctx, trace, device, conv, one_step, sync, start_events, pin, profile_report, emit_partial):
    """Returns a dict: sharded (did the group's proofs become the headline), dt / proof / phases of the group's timed region, n1, group, replica_line,
    shard_error, group_rep (the dominant kernel's records of the timed group proofs), extra_stages.
    emit_partial(extra_error): prints the strong-scaling line WITHOUT the extra stages — what the second watchdog uses when those never come back."""
    lib, dist, rank, world = pkg.lib(), R.dist, R.rank, R.world
    out = {"sharded": True, "shard_error": None, "n1": None, "group": None, "replica_line": None, "group_rep": None, "extra_stages": {}}

    def join_group(c):
        # control plane only: rank 0's RCCL unique id reaches the others through torch.distributed; every data-path exchange of the proof
        # is issued by libbfhip itself on the context's stream (RCCL over xGMI, device buffers on both ends)
        dev = R.torch.device("cuda", device) if args.dist_backend == "nccl" else None
        c.set_shard_policy(args.shard_policy)
        c.join_rccl_group(replicas.share_unique_id(dist, pkg.rccl_unique_id, dev), rank, world)

    # ---- first the N independent proofs, one per GPU, under the contract's protocol (W warm-up, barrier, K steps, barrier, MAX over ranks): (a) the
    # `replicas` field, (b) the line this run prints if the group below never comes back — a multi-GPU run always yields a line.
    R.note(f"replicas: {args.warmup} + {args.steps} proofs per GPU")
    with pin():
        dt_r, (proof_r, _) = replicas.timed_region(one_step, args.steps, args.warmup, dist=dist, sync_fn=sync, backend_tensor=R.cuda_t)
    cells_r = replicas.aggregate_units(trace.cells, dist=dist, backend_tensor=R.cuda_t)
    replica_line = {"what": "N independent proofs, one per GPU, no data-path collective (weak scaling; the headline before round 5, and with --replicas)", "value": cells_r * args.steps / dt_r,
                    "unit": "trace cells/s", "ms_per_step": dt_r / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup, "scaling": "weak", "proof_sha256": hashlib.sha256(proof_r).hexdigest()}
    out["replica_line"] = replica_line
    # ---- the one-GPU reference speedup_vs_n1 divides by: rank 0 ALONE on its GPU, the other ranks idle at a barrier (r06; before, the replicas run
    # stood in — N proofs at the same time, slowest rank, host and PCIe contention included: not "the same proof on one GPU alone")
    R.barrier()
    n1_ms, n1_sha = replica_line["ms_per_step"], replica_line["proof_sha256"]
    if rank == 0:
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            proof_1, _ = one_step()
        sync()
        n1_ms, n1_sha = (time.perf_counter() - t0) / args.steps * 1e3, hashlib.sha256(proof_1).hexdigest()
    R.barrier()
    n1_ms = R.over_ranks(n1_ms if rank == 0 else 0.0, "max")
    out["n1"] = {"ms_per_proof": n1_ms, "proof_sha256": n1_sha if rank == 0 else None, "steps": args.steps,
                 "note": "rank 0 alone on its GPU, every other rank idle at a barrier (same process, same context as the group's proofs)",
                 "replicas_ms_per_proof_slowest_rank": replica_line["ms_per_step"]}

    # ---- deployable node throughput at the metric's size: a pool of 3 per GPU (bfhip_prove_batch from one caller thread per rank), all ranks at the same time, no
    # data-path collective — weak scaling of what INTEGRATION.md section 4b deploys. Never `value`; a failure costs only this field.
    out["replicas_pool"] = replicas_pool(args, R, pkg, device)

    def group_never_came_back():
        # the group's part has not finished within --group-timeout: a collective that cannot be interrupted from here (a hung bootstrap, a wedged queue).
        # Rank 0 prints the replicas line and every rank leaves — with a NON-ZERO code (r06, ADVICE r05: a process that has touched the GPU and is stuck
        # in a collective did not succeed; bench.py's own launcher relays rank 0's line whatever the exit code, and under torchrun it is on stdout already).
        if rank == 0:
            print(json.dumps(replicas_fallback_line(args, R, trace, conv, replica_line,
                                                    f"the shard group (ONE proof over the {world} GPUs) did not finish within {args.group_timeout} s and could not be interrupted — value / "
                                                    "ms_per_step are the REPLICAS (weak scaling), measured before the group formed; exit code 3")), flush=True)
        print(f"bench.py[rank {rank}/{world}] the shard group did not come back within {args.group_timeout} s: leaving with exit code 3", file=sys.stderr, flush=True)
        if rank != 0:
            time.sleep(2.0)      # rank 0's line first: a launcher that sees a rank fail ends the others at once
        os._exit(3)

    watchdog = threading.Timer(args.group_timeout, group_never_came_back)
    watchdog.daemon = True
    watchdog.start()
    R.note(f"one-GPU reference {n1_ms:.2f} ms; joining the shard group")
    try:
        join_group(ctx)
        ok = True
        R.note("joined: " + ctx.group_info()[2])
    except Exception as e:
        ok, out["shard_error"] = False, f"joining the shard group failed on rank {rank}: {e!r}"
    if not R.agree(ok):
        out["shard_error"] = out["shard_error"] or "joining the shard group failed on another rank"
        try:
            ctx.leave_group()
        except Exception:
            pass
        watchdog.cancel()
        out["sharded"] = False
        return out

    R.note(f"timed region: {args.warmup} + {args.steps} proofs over the group")
    comm_before = {}

    def start_group_events():
        ctx.sync()                                   # group_times() wants a drained stream
        comm_before.update(ctx.group_times())
        ctx.group_latency(reset=True)                # the histogram covers the timed proofs only
        start_events()

    comm_ms, group = {}, None
    try:
        with pin():
            dt, (proof, phases) = replicas.timed_region(one_step, args.steps, args.warmup, dist=dist, sync_fn=sync, backend_tensor=R.cuda_t, on_timed_start=start_group_events)
        comm_after = ctx.group_times()
        comm_ms = {k: (comm_after[k] - comm_before.get(k, 0.0)) / args.steps for k in comm_after}
        group = {"transport": ctx.group_info()[2], "shard_policy": {"requested": args.shard_policy, "replicated_transforms": bool(ctx.last_proof_flags().get("replicated_transforms"))},
                 "per_proof_rank0": {k: round(v / (args.warmup + args.steps), 1) for k, v in ctx.group_stats().items()},
                 "comm_ms_per_proof_rank0": {k: round(v, 3) for k, v in comm_ms.items()}}
        # per kind of collective: count, p50 / p90 / max of the GPU-side duration and of the host-side call time over the timed proofs — "RCCL call
        # latency x ~31 collectives per proof" beside the bytes (the first real N-GPU run has to show which of the two decides)
        group["collective_latency_us_rank0"] = ctx.group_latency()
        out.update(dt=dt, proof=proof, phases=phases)
        ok = True
    except Exception as e:
        ok, out["shard_error"] = False, f"the shard group's proof failed on rank {rank}: {e!r}"
    R.note("group proofs done" if ok else f"group proofs FAILED: {out['shard_error']}")
    if not R.agree(ok):
        out["shard_error"] = out["shard_error"] or "the shard group's proof failed on another rank"
        out["sharded"] = False
        lib.bfhip_profile_enable(ctx._h, 0)
    else:
        # only now, with every rank known to be here: collectives of the timing channel are never issued from inside a try block a peer may have left
        group["comm_ms_per_proof_max_rank"] = round(R.over_ranks(sum(comm_ms.values()), "max"), 3)
        out["group"] = group
    try:
        ctx.leave_group()
    except Exception:
        pass
    watchdog.cancel()            # the headline group is done (or has failed cleanly): its watchdog must not fire during the extra stages
    if not out["sharded"]:
        return out
    if not args.no_kernel_events:
        out["group_rep"] = profile_report(lib, ctx)             # the dominant kernel of the TIMED group proofs (this rank's share)
        lib.bfhip_profile_enable(ctx._h, 0)

    # ---- after the headline: two proofs in flight over the shard group (on request), then BASELINE configs 3/4 literal (synthetic 2^24-row trace) and 5
    # (2^26 rows, Poseidon252) over a second group, each with its one-GPU time. Under a watchdog of their own that still emits the strong-scaling line
    # measured above (r06, ADVICE r05: the headline watchdog used to discard it).
    if args.no_extra_stages and args.group_inflight < 2:
        return out
    extra = out["extra_stages"]

    def extras_never_came_back():
        if rank == 0:
            emit_partial(out, f"the stages after the headline did not finish within {args.group_timeout} s and could not be interrupted; the headline above was measured before them; exit code 3")
        print(f"bench.py[rank {rank}/{world}] the stages after the headline did not come back within {args.group_timeout} s: leaving with exit code 3", file=sys.stderr, flush=True)
        if rank != 0:
            time.sleep(2.0)
        os._exit(3)

    watchdog2 = threading.Timer(args.group_timeout, extras_never_came_back)
    watchdog2.daemon = True
    watchdog2.start()
    if args.group_inflight >= 2:
        R.note(f"{args.group_inflight} proofs in flight over the shard group")
        extra["fib19_%d_in_flight" % args.group_inflight] = group_inflight(args, R, pkg, ctx, trace, device, join_group, hashlib.sha256(out["proof"]).hexdigest(), out["dt"] / args.steps * 1e3)
    if not args.no_extra_stages:
        R.note("extra stages: 2^24-row trace (configs 3/4), 2^26-row Poseidon252 trace (config 5)")
        big = None
        try:
            big = pkg.Context(device, max_log_domain=28)
            ok = True
        except Exception as e:
            ok = False
            extra["error"] = f"rank {rank}: creating the 2^28 context failed: {e!r}"
        # agreed BEFORE anyone enters share_unique_id's broadcast: a rank whose context creation failed must not skip a collective the others are in
        if R.agree(ok):
            try:
                stages = [("trace_2p24_blake2s", sweep_program(24), 24, (0, 0, 0, 0), 1, 3, None), ("trace_2p26_poseidon252", sweep_program(26), 26, (0, 0, 0, 1), 1, 1, None)]
                join_group(big)
                probe_run_stages(pkg, [big], stages, {"stages": extra}, lambda: None, rank == 0, ref_device=device, max_log=28)
                ok = True
            except Exception as e:
                ok = False
                extra["error"] = f"rank {rank}: {e!r}"
            if not R.agree(ok):
                extra.setdefault("error", "a stage failed on another rank")
        else:
            extra.setdefault("error", "creating the 2^28 context failed on another rank")
        if big is not None:
            try:
                big.leave_group()
            except Exception:
                pass
            big.close()
    watchdog2.cancel()
    return out


def replicas_pool(args, R, pkg, device, log=22, k=3, batch=12, batches=3):
    """Every rank proves batches of the synthetic 2^22-row trace (the metric's size) through its own pool of k sub-contexts; barrier, timed batches, barrier, MAX
    over ranks; value = cells of all ranks / time."""
    row = {"what": f"N pools of {k} (one per GPU, one caller thread each): batches of {batch} proofs of the synthetic 2^{log}-row trace, all ranks at once, no data-path collective",
           "in_flight_per_gpu": k, "scaling": "weak"}
    pool = tr = None
    ok, dt, cells, sha = True, 0.0, 0, None
    try:
        pool = pkg.Pool(device, n_in_flight=k, max_log_domain=log + 2)
        tr = pkg.Trace(pool.ctx(0), sweep_program(log), b"")
        traces, cells = [tr] * batch, tr.cells
        pool.prove_batch(traces, log, want_json=False)
    except Exception as e:
        ok, row["error"] = False, f"rank {R.rank}: {e!r}"
    if R.agree(ok):
        try:
            R.barrier()
            t0 = time.perf_counter()
            for _ in range(batches):
                pool.prove_batch(traces, log, want_json=False)
            dt = time.perf_counter() - t0
            sha = hashlib.sha256(pool.prove_batch(traces[:1], log)[0][0]).hexdigest()
        except Exception as e:
            ok, row["error"] = False, f"rank {R.rank}: {e!r}"
        dt = R.over_ranks(dt, "max")
        if R.agree(ok) and dt > 0:
            total = R.over_ranks(cells * batch * batches, "sum")
            row.update(value=total / dt, unit="trace cells/s", ms_per_proof_per_gpu=round(dt / (batch * batches) * 1e3, 3), cells_per_s_per_gpu=round(total / dt / R.world), proof_sha256=sha)
        else:
            row.setdefault("error", "a batch failed on another rank")
    else:
        row.setdefault("error", "creating the pool failed on another rank")
    try:
        if tr is not None:
            tr.close()
        if pool is not None:
            pool.close()
    except Exception:
        pass
    return row


def group_inflight(args, R, pkg, ctx, trace, device, join_group, want_sha, ms_one_in_flight):
    """--group-inflight K: K proofs in flight over the shard group — every rank drives K contexts (K host threads), context j of every rank forms group j
    (its own communicator), and the K groups prove the bench workload at the same time: the share of a sharded proof that every rank repeats (tree tops, the
    FRI small end: S = 3.3 of 29 ms in the one-GPU projection) and the waits inside the collectives of one group are filled by the
    other group's divided work. ms_per_proof = wall time / (K x proofs per group). Opt-in: two communicators driven concurrently from two threads are
    exercised here through the process-per-rank double (tests/mock_rccl_ipc.cpp); with librccl itself this is unmeasured."""
    k = args.group_inflight
    row = {"in_flight": k, "what": f"{k} shard groups at the same time, one context and one host thread per group and rank; ms_per_proof = wall time / proofs completed"}
    ctxs, traces, ok = [ctx], [trace], True
    try:
        for _ in range(k - 1):
            c2 = pkg.Context(device, max_log_domain=args.log_max_rows + 2)
            ctxs.append(c2)
            traces.append(pkg.Trace(c2, FIB19, b""))
    except Exception as e:
        ok, row["error"] = False, f"rank {R.rank}: {e!r}"
    if R.agree(ok):
        joined = []
        try:
            for c in ctxs:                      # the same order on every rank: one unique id per group over the timing channel
                join_group(c)
                joined.append(c)
            ok = True
        except Exception as e:
            ok, row["error"] = False, f"rank {R.rank}: joining group {len(joined)} failed: {e!r}"
        if R.agree(ok):
            res, errs = [None] * k, []

            def run(j, n):
                try:
                    for _ in range(n):
                        res[j] = traces[j].prove(args.log_max_rows)[0]
                    ctxs[j].sync()
                except Exception as e:
                    errs.append(repr(e))

            def wave(n):
                th = [threading.Thread(target=run, args=(j, n)) for j in range(k)]
                [t.start() for t in th]; [t.join() for t in th]

            wave(1)                             # warm-up (the second context's arena, the groups' first collectives)
            R.barrier()
            t0 = time.perf_counter()
            wave(args.steps)
            dt = R.over_ranks(time.perf_counter() - t0, "max") if not errs else 0.0
            ok = not errs
            if errs:
                row["error"] = f"rank {R.rank}: " + "; ".join(errs)
            if R.agree(ok):
                ms = dt / (k * args.steps) * 1e3
                shas = [hashlib.sha256(p).hexdigest() for p in res]
                row.update(ms_per_proof=round(ms, 3), cells_per_s=trace.cells / (ms * 1e-3), proofs_timed=k * args.steps, proof_sha256=shas[0],
                           identical_to_the_headline_proof=all(s == want_sha for s in shas), ms_per_proof_one_in_flight=round(ms_one_in_flight, 3),
                           gain_vs_one_in_flight=round(ms_one_in_flight / ms, 3), transport=ctxs[0].group_info()[2])
            else:
                row.setdefault("error", "a proof failed on another rank")
        else:
            row.setdefault("error", "joining a group failed on another rank")
        for c in joined:
            try:
                c.leave_group()
            except Exception:
                pass
    else:
        row.setdefault("error", "creating the extra contexts failed on another rank")
    for t2, c2 in zip(traces[1:], ctxs[1:]):
        try:
            t2.close(); c2.close()
        except Exception:
            pass
    return row
