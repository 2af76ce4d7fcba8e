#!/usr/bin/env python3
"""Microbenchmark of k_merkle_layer per layer shape through the C ABI (bfhip_merkle_commit_layer): compressions/s against the
measured Blake2s peak (39.9 G/s, tools/ubench_blake.hip). Shapes are the ones a fib19 proof is made of."""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("stwo_brainfuck_amd", os.path.join(ROOT, "stwo-brainfuck_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec); sys.modules["stwo_brainfuck_amd"] = pkg; spec.loader.exec_module(pkg)

PEAK = 39.9e9


def run(ctx, log, has_prev, ncols, shift=0, reps=30):
    n = 1 << log
    cols = [ctx.malloc(4 * max(1, n >> shift)) for _ in range(ncols)]
    for p in cols:
        pkg._check(pkg.lib().bfhip_memset_zero(ctx._h, __import__("ctypes").c_void_p(p), __import__("ctypes").c_size_t(4 * max(1, n >> shift))))
    prev = ctx.malloc(64 * n) if has_prev else 0
    out = ctx.malloc(32 * n)
    sh = [shift] * ncols
    ctx.merkle_commit_layer(log, prev, cols, out, sh); ctx.sync()
    t = time.perf_counter()
    for _ in range(reps):
        ctx.merkle_commit_layer(log, prev, cols, out, sh)
    ctx.sync()
    dt = (time.perf_counter() - t) / reps
    msg = (64 if has_prev else 0) + 4 * ncols
    blocks = max(1, -(-msg // 64))
    comp = n * blocks
    for p in cols + [out] + ([prev] if prev else []):
        ctx.free(p)
    return dt, comp


if __name__ == "__main__":
    ctx = pkg.Context(0, max_log_domain=16)
    rows = []
    for name, log, prev, ncols, shift in [
        ("leaf 4 cols (composition / FRI leaf)", 25, False, 4, 0), ("leaf 4 cols", 22, False, 4, 0), ("leaf 1 col (IsFirst)", 25, False, 1, 0),
        ("inner, no cols", 24, True, 0, 0), ("inner, no cols", 22, True, 0, 0), ("inner, no cols", 21, True, 0, 0), ("inner, no cols", 20, True, 0, 0), ("inner, no cols", 19, True, 0, 0), ("inner, no cols", 18, True, 0, 0), ("inner, no cols", 16, True, 0, 0), ("inner, no cols", 13, True, 0, 0),
        ("inner + 1 col (IsFirst)", 24, True, 1, 0), ("inner + 4 cols (FRI first layer)", 24, True, 4, 0), ("inner + 16 cols", 22, True, 16, 0),
        ("leaf 16 cols", 24, False, 16, 0), ("leaf 64 cols", 22, False, 64, 0), ("leaf 128 cols", 21, False, 128, 0),
    ]:
        dt, comp = run(ctx, log, prev, ncols, shift)
        rows.append((name, log, comp, dt))
        print(f"{name:40s} log {log:2d}  {comp/1e6:8.1f} M compressions  {dt*1e6:9.1f} us  {comp/dt/1e9:6.2f} G/s  {comp/dt/PEAK*100:5.1f} % of peak")
    ctx.close()
