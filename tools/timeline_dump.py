#!/usr/bin/env python3
"""Lists every kernel of the LAST proof in a rocprofv3 --kernel-trace CSV: start offset, duration, idle gap in front of it, grid size,
stream/queue. Usage: timeline_dump.py <kernel_trace.csv> [--summary]
--summary: per kernel name {launches, total us, launches shorter than 10 us and their total} — where the fixed cost of a small proof sits."""
import csv
import sys


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bf::", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0),
                         int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0), r.get("Queue_Id", "")))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_is_first_coeffs") and (i == 0 or not rows[i - 1][2].startswith("k_is_first_coeffs"))]
    return rows[starts[-1] if starts else 0:]


def main():
    rows = load(sys.argv[1])
    t0 = rows[0][0]
    if "--summary" in sys.argv:
        agg = {}
        for s, e, name, g, w, q in rows:
            a = agg.setdefault(name, [0, 0, 0, 0])
            a[0] += 1; a[1] += e - s
            if e - s < 10000:
                a[2] += 1; a[3] += e - s
        span = max(r[1] for r in rows) - t0
        print(f"span {span/1e3:.1f} us, {len(rows)} launches, sum of kernel durations {sum(a[1] for a in agg.values())/1e3:.1f} us")
        print(f"{'kernel':44s} {'n':>5s} {'total us':>10s} {'n<10us':>7s} {'their us':>9s}")
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"{name:44s} {a[0]:5d} {a[1]/1e3:10.1f} {a[2]:7d} {a[3]/1e3:9.1f}")
        return
    end_prev = {}
    last_end = t0
    for s, e, name, g, w, q in rows:
        gap = s - last_end
        print(f"+{(s-t0)/1e3:9.1f} us  {(e-s)/1e3:8.1f} us  gap {gap/1e3:7.1f}  q{q:>3s} grid {g:>9d} wg {w:>4d}  {name}")
        last_end = max(last_end, e)


if __name__ == "__main__":
    main()
