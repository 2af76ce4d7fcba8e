#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) and reports, for the LAST proof in the trace, the GPU busy fraction and
the largest idle gaps between consecutive kernels (with the kernels on either side). A k_mailbox launch (waiting for the host) counts as idle. Usage: timeline_gaps.py <kernel_trace.csv> [n_gaps]
--window A:B  (fractions of the trace's span, e.g. 0.45:0.85) instead of the last proof: every queue's launches inside that window — for k proofs in
flight (tools/inflight_profile.py): busy fraction, the share of the time with launches of two or more QUEUES in flight (one proof's latency chain
under another proof's kernels), proofs started in the window."""
import csv
import sys


def main():
    path = sys.argv[1]
    window = None
    argv = list(sys.argv[2:])
    if "--window" in argv:
        i = argv.index("--window"); a, b = argv[i + 1].split(":"); window = (float(a), float(b)); del argv[i:i + 2]
    n_gaps = int(argv[0]) if argv else 25
    rows, queues = [], []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bf::", "")))
            queues.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), rows[-1][2]))
    rows.sort()
    if window:
        lo_t, hi_t = rows[0][0], max(r[1] for r in rows)
        w0, w1 = lo_t + window[0] * (hi_t - lo_t), lo_t + window[1] * (hi_t - lo_t)
        inside = sorted(q for q in queues if q[0] >= w0 and q[1] <= w1 and not q[3].startswith("k_mailbox"))
        # sweep over start/end events: time with >= 1 launch in flight, and with launches of >= 2 different queues in flight
        ev = []
        for s_, e_, q, _ in inside:
            ev.append((s_, 1, q)); ev.append((e_, -1, q))
        ev.sort()
        active, busy, multi, last = {}, 0, 0, None
        for t, d, q in ev:
            if last is not None:
                nq = sum(1 for v in active.values() if v > 0)
                if nq >= 1: busy += t - last
                if nq >= 2: multi += t - last
            active[q] = active.get(q, 0) + d
            last = t
        span = w1 - w0
        starts = sum(1 for i, q in enumerate(inside) if q[3].startswith("k_is_first_coeffs"))
        per_q = {}
        for s_, e_, q, _ in inside:
            per_q.setdefault(q, [0, 0]); per_q[q][0] += 1; per_q[q][1] += e_ - s_
        print(f"window {window[0]:.2f}..{window[1]:.2f} of the trace: {span/1e6:.3f} ms, {len(inside)} launches on {len(per_q)} queues, {starts} proofs started")
        print(f"busy (>= 1 launch in flight) {busy/1e6:.3f} ms = {100.0*busy/span:.1f} %   idle {(span-busy)/1e6:.3f} ms")
        print(f"launches of >= 2 queues in flight {multi/1e6:.3f} ms = {100.0*multi/span:.1f} % of the window (one proof's chains under another's kernels)")
        if starts:
            print(f"window / proofs started = {span/1e6/starts:.3f} ms per proof")
        for q, (n, tot) in sorted(per_q.items(), key=lambda kv: -kv[1][1]):
            print(f"  queue {q}: {n} launches, {tot/1e6:.3f} ms of kernel time")
        return
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
