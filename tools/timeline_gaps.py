#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) and reports, for the LAST proof in the trace, the GPU busy fraction and
the largest idle gaps between consecutive kernels (with the kernels on either side). A k_mailbox launch (waiting for the host) counts as idle. Usage: timeline_gaps.py <kernel_trace.csv> [n_gaps]"""
import csv
import sys


def main():
    path = sys.argv[1]
    n_gaps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bf::", "")))
    rows.sort()
    # k_mailbox (csrc/mailbox.hip) is one workgroup waiting for the host's flag: its duration is idle time of the GPU, not work — it is reported
    # as a gap in front of the kernel that follows it (the few microseconds of its table copy are counted as idle too)
    mailbox = [r for r in rows if r[2].startswith("k_mailbox")]
    rows = [r for r in rows if not r[2].startswith("k_mailbox")]
    # a proof starts with the k_is_first_coeffs burst of the preprocessed phase: take the last such burst as the start of the last proof
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_is_first_coeffs") and (i == 0 or not rows[i - 1][2].startswith("k_is_first_coeffs"))]
    first = starts[-1] if starts else 0
    rows = rows[first:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    # union of intervals (two streams may overlap)
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    gaps = []
    for s, e, name in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e - t0, name))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = t1 - t0
    if mailbox:
        inside = [m for m in mailbox if m[0] >= t0]
        print(f"k_mailbox: {len(inside)} launches, {sum(m[1] - m[0] for m in inside)/1e3:.1f} us waiting for the host (counted as idle below)")
    print(f"launches {len(rows)}  span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms  ({100.0*busy/span:.1f} %)  idle {(span-busy)/1e6:.3f} ms in {len(gaps)} gaps")
    hist = {}
    for g, at, name in gaps:
        hist.setdefault(name, [0, 0])
        hist[name][0] += 1; hist[name][1] += g
    print("idle time by the kernel that follows the gap:")
    for name, (cnt, tot) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"  {name:40s} {cnt:4d} gaps  {tot/1e3:8.1f} us")
    print(f"largest {n_gaps} gaps:")
    for g, at, name in sorted(gaps, reverse=True)[:n_gaps]:
        print(f"  {g/1e3:8.1f} us at +{at/1e6:7.3f} ms before {name}")


if __name__ == "__main__":
    main()
