#!/usr/bin/env python3
"""GPU idle time of ONE proof by N ranks time-sharing a GPU (rocprofv3 --kernel-trace CSV of tools/shard_kernels.py + its JSON line):
union of all ranks' kernel intervals over the last proof, the idle gaps with the launches on either side and the thread rocprofv3 names for them (an approximation: see below),
and the idle time attributed to the kernel that ends each gap. With every rank's work on one device the wall time of a proof is
(sum of kernel time) - (overlap of co-running launches) + (idle): the idle part is host latency (Fiat-Shamir round trips, rendezvous of the
in-process transport), not work, and does not scale with N the way replicated work does.
Usage: shard_timeline.py <kernel_trace.csv> <run.json> [n_gaps]"""
import csv
import json
import sys


def main():
    path, run = sys.argv[1], json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    n_gaps = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    rank_of = {tid: k for k, tid in enumerate(run["rank_thread_ids"])}
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            tid = int(r["Thread_Id"])
            if tid not in rank_of:
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bf::", ""), rank_of[tid]))
    rows.sort()
    # every rank starts a proof with one k_is_first_coeffs launch: the last proof begins at the N-th such launch from the end (rocprofv3 does
    # not attribute every dispatch of a multi-threaded process to the thread that issued it, so this is counted, not looked up per rank)
    firsts = [s for s, e, name, rk in rows if name.startswith("k_is_first_coeffs")]
    t0 = firsts[-run["ranks_on_one_gpu"]]
    rows = [r for r in rows if r[0] >= t0]
    t1 = max(r[1] for r in rows)
    busy, cur_s, cur_e, last = 0, rows[0][0], rows[0][1], rows[0]
    gaps = []
    for row in rows[1:]:
        s, e, name, rk = row
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e - t0, last, row))
            cur_s, cur_e, last = s, e, row
        elif e > cur_e:
            cur_e, last = e, row
    busy += cur_e - cur_s
    span = t1 - t0
    ksum = sum(e - s for s, e, _, _ in rows)
    print(f"{run['workload']}, {run['ranks_on_one_gpu']} ranks on one GPU: last proof span {span / 1e6:.3f} ms, GPU busy (union) {busy / 1e6:.3f} ms = {100.0 * busy / span:.1f} %, "
          f"idle {(span - busy) / 1e6:.3f} ms in {len(gaps)} gaps; sum of kernel durations {ksum / 1e6:.3f} ms ({len(rows)} launches)")
    hist = {}
    for g, at, before, after in gaps:
        h = hist.setdefault(after[2], [0, 0])
        h[0] += 1; h[1] += g
    print("idle time by the kernel that ends the gap:")
    for name, (cnt, tot) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {name[:48]:48s} {cnt:4d} gaps {tot / 1e3:9.1f} us")
    print(f"largest {n_gaps} gaps:")
    for g, at, before, after in sorted(gaps, key=lambda x: -x[0])[:n_gaps]:
        print(f"  {g / 1e3:8.1f} us at +{at / 1e6:7.3f} ms   after {before[2][:30]:30s} (rank {before[3]})   before {after[2][:30]:30s} (rank {after[3]})")


if __name__ == "__main__":
    main()
