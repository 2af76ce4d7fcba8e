#!/usr/bin/env python3
"""One proof over N contexts of ONE GPU (local shard group, one host thread per rank): ms per proof for N = 1, 2, 4, 8 and what it says
about the division of work. The N ranks time-share the GPU, so T(N) = N * S + P where S is the work every rank repeats (replicated
phases) and P the work that is divided; with one GPU per rank the expected time is about S + P / N plus the exchanges.
Usage: python tools/shard_local.py [steps]"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package

FIB19 = open(os.path.join(ROOT, "tests", "golden", "programs", "fib19.bf")).read()
# optional: SHARD_LOCAL_POSEIDON=1 proves with the Poseidon252MerkleChannel variant, SHARD_LOCAL_LOG=k uses the synthetic 2^k-row trace of
# bench.py's sweep with LOG_MAX_ROWS = k (BASELINE config 5 family) instead of fib19
POSEIDON = os.environ.get("SHARD_LOCAL_POSEIDON") == "1"
SYN_LOG = int(os.environ.get("SHARD_LOCAL_LOG", "0"))
OVERLAP = int(os.environ.get("SHARD_LOCAL_OVERLAP", "0"))      # bfhip_ctx_set_overlap mask of every rank (4: exchanges on the partner stream)
POLICY = int(os.environ.get("SHARD_LOCAL_POLICY", "0"))        # bfhip_ctx_set_shard_policy of every rank: 0 exchange columns -> rows, 1 replicate the transforms


def run(pkg, n, steps, lmr=24):
    global FIB19
    if SYN_LOG:
        lmr = SYN_LOG
        FIB19 = "+" * 14 + "[>" + "+" * (250 << (SYN_LOG - 20)) + "[>+<-]<-]"
    pkg.set_default_conventions(0, 0, 0, 1 if POSEIDON else 0)
    group = pkg.LocalGroup(n) if n > 1 else None
    ctxs = [pkg.Context(0, max_log_domain=lmr + 2) for _ in range(n)]
    traces = [pkg.Trace(c, FIB19, b"") for c in ctxs]
    proofs, stats, times, phases = [None] * n, [None] * n, [0.0] * n, [None] * n
    barrier = threading.Barrier(n)
    step_times = [[0.0] * steps for _ in range(n)]

    def work(r):
        if group:
            ctxs[r].join_local_group(group, r)
            if OVERLAP:
                ctxs[r].set_overlap(OVERLAP)
            ctxs[r].set_shard_policy(POLICY)
        traces[r].prove(lmr)
        # the first proof of a context loads code objects (seconds, inside whatever collective comes first): counters and collective times are
        # taken as differences over the timed proofs only
        base = (ctxs[r].group_stats(), ctxs[r].group_times()) if group else None
        barrier.wait()
        t0 = time.perf_counter()
        for k in range(steps):
            tk = time.perf_counter()
            proofs[r], phases[r] = traces[r].prove(lmr)
            step_times[r][k] = time.perf_counter() - tk
        ctxs[r].sync()
        times[r] = (time.perf_counter() - t0) / steps
        if group:
            after, t_after = ctxs[r].group_stats(), ctxs[r].group_times()
            stats[r] = {k: after[k] - base[0][k] for k in after}
            stats[r]["times_ms"] = {k: t_after[k] - base[1][k] for k in t_after}
    th = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    [t.start() for t in th]; [t.join() for t in th]
    for r in range(n):
        if group:
            ctxs[r].leave_group()
        traces[r].close(); ctxs[r].close()
    if group:
        group.close()
    assert all(p == proofs[0] for p in proofs)
    # the fastest proof: N host threads of one process time-sharing one GPU are at the mercy of the host's scheduler (box to box the mean of
    # T(8) moves by 10 %); a proof is done when its slowest rank is
    fastest = min(max(step_times[r][k] for r in range(n)) for k in range(steps))
    return max(times) * 1e3, proofs[0], stats, phases[0], fastest * 1e3


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    pkg = load_package()
    out, ref = [], None
    for n in (1, 2, 4, 8):
        ms, proof, stats, ph, fastest = run(pkg, n, steps)
        ref = ref or proof
        row = {"ranks_on_one_gpu": n, "ms_per_proof": round(ms, 2), "ms_fastest_proof": round(fastest, 2), "identical_to_single": proof == ref,
               "rank0_phase_ms_last_proof": {k: round(v * 1e3, 2) for k, v in ph.items()}}
        if stats[0]:
            per_proof = {k: v / steps for k, v in stats[0].items() if k != "times_ms"}
            # GPU-side time inside the collectives, per proof (HIP-event pairs; on one shared GPU this includes waiting for the peers' kernels)
            row["rank0_collective_ms_per_proof"] = {k: round(v / steps, 3) for k, v in stats[0]["times_ms"].items()}
            row["rank0_per_proof"] = {"all_gathers": per_proof["all_gathers"], "max_reduces": per_proof["max_reduces"], "exchanges": per_proof["exchanges"],
                                      "MB_sent": round(per_proof["bytes_sent"] / 1e6, 1)}
        out.append(row)
    t1 = out[0]["ms_per_proof"]
    p1 = out[0]["rank0_phase_ms_last_proof"]
    for row in out[1:]:
        n = row["ranks_on_one_gpu"]
        # per phase: T(N) = N S + P with T(1) = S + P  ->  S = (T(N) - T(1)) / (N - 1): the part of the phase every rank repeats
        row["replicated_ms_by_phase"] = {k: round((row["rank0_phase_ms_last_proof"][k] - p1[k]) / (n - 1), 2) for k in p1}
    for row in out[1:]:
        n = row["ranks_on_one_gpu"]
        s = (row["ms_per_proof"] - t1) / (n - 1)          # T(N) = N S + P, T(1) = S + P
        row["replicated_ms_S"] = round(s, 2); row["divided_ms_P"] = round(t1 - s, 2); row["projected_ms_with_one_gpu_per_rank"] = round(s + (t1 - s) / n, 2)
        f1 = out[0]["ms_fastest_proof"]; sf = (row["ms_fastest_proof"] - f1) / (n - 1)
        row["from_the_fastest_proofs"] = {"replicated_ms_S": round(sf, 2), "divided_ms_P": round(f1 - sf, 2), "projected_ms_with_one_gpu_per_rank": round(sf + (f1 - sf) / n, 2)}
    print(json.dumps({"workload": (f"synthetic 2^{SYN_LOG}-row trace" if SYN_LOG else "fib19.bf") + (", Poseidon252MerkleChannel" if POSEIDON else ", Blake2sMerkleChannel"),
                      "shard_policy": "replicate the transforms (1)" if POLICY == 1 else "exchange columns -> rows (0)",
                      "runs": out}, indent=1))


if __name__ == "__main__":
    main()
