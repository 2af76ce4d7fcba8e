#!/usr/bin/env python3
"""What does every rank of a shard group repeat? Reads two rocprofv3 --kernel-trace CSVs of tools/shard_kernels.py — one rank, and N ranks
time-sharing one GPU — with the JSON line each run printed, and tabulates per kernel name:

  one        GPU time per proof with one rank (us)
  sum_N      GPU time per proof summed over the N ranks (us)            -> sum_N - one = work the group adds (redundant or overhead)
  per_rank   sum_N / N = what one GPU of an N-GPU group would execute    (mean over ranks; max over ranks beside it)
  ideal      one / N

Kernel durations on a shared GPU are stretched when launches of different ranks co-run, so sum_N is an upper estimate of the work; kernels
of the copy engine / blit kernels (__amd_rocclr_*) stand for the in-process transport's device-to-device copies.

Usage: shard_redundancy.py <trace_1.csv> <run_1.json> <trace_N.csv> <run_N.json>"""
import csv
import json
import sys


def short(name):
    name = name.split("(")[0].replace("void ", "").replace("bf::", "")
    return name


def load(path, with_grid=False):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            row = (int(r["Thread_Id"]), short(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            rows.append(row + (int(r.get("Grid_Size_X", 0) or 0) // max(int(r.get("Workgroup_Size_X", 1) or 1), 1),) if with_grid else row)
    return rows


def by_launch_size(rows, tids, proofs, kernel):
    """{log2(workgroups) bucket: [launches per proof, us per proof]} of one kernel over the given threads"""
    h = {}
    for tid, name, d, wgs in rows:
        if tid in tids and name == kernel:
            b = max(wgs, 1).bit_length() - 1
            e = h.setdefault(b, [0.0, 0.0])
            e[0] += 1.0 / proofs; e[1] += d / 1e3 / proofs
    return h


def main():
    t1, j1, tn, jn = sys.argv[1:5]
    r1 = json.loads(open(j1).read().strip().splitlines()[-1])
    rn = json.loads(open(jn).read().strip().splitlines()[-1])
    n = rn["ranks_on_one_gpu"]
    p1, pn = r1["proofs_in_trace_per_rank"], rn["proofs_in_trace_per_rank"]
    one = {}
    tid1 = set(r1["rank_thread_ids"])
    for tid, name, d in load(t1):
        if tid in tid1:
            one[name] = one.get(name, 0) + d / p1
    rank_of = {tid: k for k, tid in enumerate(rn["rank_thread_ids"])}
    per = {}
    cnt = {}
    for tid, name, d in load(tn):
        if tid not in rank_of:
            continue
        per.setdefault(name, [0.0] * n)[rank_of[tid]] += d / pn
        cnt[name] = cnt.get(name, 0) + 1
    names = sorted(set(one) | set(per), key=lambda k: -(sum(per.get(k, [0])) - one.get(k, 0)))
    tot_one = sum(one.values()) / 1e6
    tot_sum = sum(sum(v) for v in per.values()) / 1e6
    rank_tot = [sum(v[k] for v in per.values()) / 1e6 for k in range(n)]
    print(f"workload: {rn['workload']}; N = {n}; proofs per rank in the traces: {p1} / {pn}; wall ms per proof: {r1['ms_per_proof_wall']} (1 rank) / {rn['ms_per_proof_wall']} ({n} ranks on one GPU)")
    print(f"GPU time per proof: one rank {tot_one:.3f} ms; {n} ranks summed {tot_sum:.3f} ms = {tot_sum / n:.3f} ms per rank (max rank {max(rank_tot):.3f}); ideal {tot_one / n:.3f} ms per rank")
    print(f"added by the group (sum_N - one): {tot_sum - tot_one:.3f} ms per proof = {(tot_sum - tot_one) / max(n - 1, 1):.3f} ms per extra rank")
    print(f"{'kernel':46s} {'one us':>9s} {'sum_N us':>10s} {'added us':>9s} {'per_rank':>9s} {'max_rank':>9s} {'ideal':>8s} {'launches/proof/rank':>10s}")
    for k in names:
        o = one.get(k, 0.0) / 1e3
        v = [x / 1e3 for x in per.get(k, [0.0] * n)]
        s = sum(v)
        if o < 0.5 and s < 0.5:
            continue
        print(f"{k[:46]:46s} {o:9.1f} {s:10.1f} {s - o:9.1f} {s / n:9.1f} {max(v):9.1f} {o / n:8.1f} {cnt.get(k, 0) / pn / n:10.1f}")
    print("\nGPU time per rank and proof (ms): " + " ".join(f"{x:.2f}" for x in rank_tot))
    for k in names[:8]:
        print(f"  {k[:44]:44s} per rank (us): " + " ".join(f"{x / 1e3:.0f}" for x in per.get(k, [0.0] * n)))
    # where a kernel's added time sits: launches by size (log2 of the workgroup count), one rank vs the N ranks together
    g1, gn = load(t1, True), load(tn, True)
    for kernel in [k for k in names[:6] if not k.startswith("__amd")]:
        h1, hn = by_launch_size(g1, tid1, p1, kernel), by_launch_size(gn, set(rank_of), pn, kernel)
        print(f"\n{kernel}: launches by log2(workgroups) -> one rank: launches / us per proof | {n} ranks: launches / us per proof")
        for b in sorted(set(h1) | set(hn)):
            a, c = h1.get(b, [0, 0]), hn.get(b, [0, 0])
            print(f"  2^{b:<2d} {a[0]:7.1f} {a[1]:9.1f}   | {c[0]:7.1f} {c[1]:9.1f}")


if __name__ == "__main__":
    main()
