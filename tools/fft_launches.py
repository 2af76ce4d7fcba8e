#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and lists, for the LAST proof in the trace, every circle-FFT launch with its grid and duration,
then totals per (kernel, grid). Usage: fft_launches.py <kernel_trace.csv> [kernel name prefix = k_fft]"""
import csv
import sys
from collections import defaultdict


def main():
    prefix = sys.argv[2] if len(sys.argv) > 2 else "k_fft"
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bf::", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0),
                         int(r.get("Grid_Size_Y", 1) or 1), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_is_first_coeffs") and (i == 0 or not rows[i - 1][2].startswith("k_is_first_coeffs"))]
    rows = rows[starts[-1]:] if starts else rows
    agg = defaultdict(lambda: [0, 0.0])
    tot = 0.0
    for s, e, name, gx, gy, wx in rows:
        if not name.startswith(prefix):
            continue
        us = (e - s) / 1e3
        tot += us
        key = (name, gx // max(wx, 1), gy)
        agg[key][0] += 1; agg[key][1] += us
    print(f"{prefix}* kernels of the last proof: {tot / 1e3:.3f} ms")
    for (name, bx, by), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:28s} blocks {bx:6d} x {by:3d}  launches {n:3d}  total {us:9.1f} us  avg {us / n:8.1f} us")


if __name__ == "__main__":
    main()
