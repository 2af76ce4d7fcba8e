#!/usr/bin/env python3
"""Proves the synthetic nested-counter trace of 2^log domain rows (bench.py's sweep family) a few times — the command to put under
rocprofv3 --kernel-trace for a per-size timeline (tools/timeline_gaps.py, tools/fft_launches.py). Usage: size_profile.py [log=22] [steps=3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package

log = int(sys.argv[1]) if len(sys.argv) > 1 else 22
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pkg = load_package()
c = pkg.Context(0, max_log_domain=log + 2)
tr = pkg.Trace(c, "+" * 14 + "[>" + "+" * (250 << (log - 20)) + "[>+<-]<-]", b"")
tr.prove(log)
c.sync(); t0 = time.perf_counter()
for _ in range(steps):
    proof, ph = tr.prove(log)
c.sync()
print(f"2^{log}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms per proof", {k: round(v * 1e3, 2) for k, v in ph.items()})
tr.close(); c.close()
