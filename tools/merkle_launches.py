#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and, for the LAST proof, aggregates the Merkle layer launches by grid size: launches, kernel
duration, and start-to-start interval to the next launch (what a layer really costs on the timeline). Usage: merkle_launches.py <csv>"""
import csv, sys
from collections import defaultdict
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bf::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_one_hot") and (i == 0 or not rows[i - 1][2].startswith("k_one_hot"))]
rows = rows[starts[-1]:] if starts else rows
agg = defaultdict(lambda: [0, 0.0, 0.0])
for i, (s, e, name, blocks) in enumerate(rows):
    if not name.startswith("k_merkle"):
        continue
    nxt = rows[i + 1][0] if i + 1 < len(rows) else e
    a = agg[(name, blocks)]
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] += (nxt - s) / 1e3
tot_d = sum(v[1] for v in agg.values()); tot_i = sum(v[2] for v in agg.values())
print(f"Merkle kernels of the last proof: {tot_d / 1e3:.3f} ms of kernel time, {tot_i / 1e3:.3f} ms start-to-next-start")
small_d = sum(v[1] for (n, b), v in agg.items() if n == "k_merkle_layer" and b <= 1024); small_i = sum(v[2] for (n, b), v in agg.items() if n == "k_merkle_layer" and b <= 1024)
print(f"layers of <= 1024 blocks (<= 2^18 nodes): {small_d / 1e3:.3f} ms kernel time, {small_i / 1e3:.3f} ms on the timeline")
for (name, blocks), (n, d, iv) in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    print(f"{name:16s} blocks {blocks:7d}  launches {n:3d}  avg duration {d / n:8.1f} us  avg start-to-next {iv / n:8.1f} us")
