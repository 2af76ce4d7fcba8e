#!/usr/bin/env python3
"""Does a process's HISTORY of contexts change what two proofs in flight gain? One process: (A) two fresh contexts, 1 and 2 in flight; close; (B) the same again with two
new contexts; (C) with two idle contexts kept alive beside them. 2^22-row trace. Prints one line per phase. (profiles/r05_inflight_history.txt)"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = bench.load_package()
code, lmr = bench.sweep_program(22), 22


def measure(k, keep_alive=()):
    ctxs = [pkg.Context(0, max_log_domain=lmr + 2) for _ in range(k)]
    traces = [pkg.Trace(c, code, b"") for c in ctxs]
    def wave(n):
        th = [threading.Thread(target=lambda t=t: [t.prove(lmr, want_json=False) for _ in range(n)]) for t in traces]
        [t.start() for t in th]; [t.join() for t in th]
        for c in ctxs: c.sync()
    wave(3)
    waves, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        wave(8); waves += 1
    ms = (time.perf_counter() - t0) / (waves * 8 * k) * 1e3
    for t in traces: t.close()
    for c in ctxs: c.close()
    return round(ms, 3)


for phase in ("A fresh", "B after closed contexts", "C after more closed contexts"):
    one, two = measure(1), measure(2)
    print(f"{phase:32s} 1 in flight {one} ms, 2 in flight {two} ms per proof  (x{one / two:.3f})", flush=True)
idle = [pkg.Context(0, max_log_domain=lmr + 2) for _ in range(2)]
one, two = measure(1), measure(2)
print(f"{'D two idle contexts alive':32s} 1 in flight {one} ms, 2 in flight {two} ms per proof  (x{one / two:.3f})", flush=True)
tr = [pkg.Trace(c, code, b"") for c in idle]
for t in tr: t.prove(lmr, want_json=False)
one, two = measure(1), measure(2)
print(f"{'E those two have proved once':32s} 1 in flight {one} ms, 2 in flight {two} ms per proof  (x{one / two:.3f})", flush=True)
