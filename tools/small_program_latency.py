#!/usr/bin/env python3
"""Latency of small programs at the reference's fixed LOG_MAX_ROWS = 24 (BASELINE config 1 program and friends): the preprocessed
IsFirst tree (2^24 rows) dominates unless it is kept across proofs (bfhip_ctx_reuse_preprocessed)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package

pkg = load_package()
ctx = pkg.Context(0, max_log_domain=26)
for name, inp in [("hello_kakarot.bf", b""), ("collatz.bf", b"7\n"), ("a-bc.bf", b"a")]:
    code = open(os.path.join(ROOT, "tests", "golden", "programs", name)).read()
    for reuse in (0, 1):
        pkg.lib().bfhip_ctx_reuse_preprocessed(ctx._h, reuse)
        for _ in range(3):
            pkg.prove_brainfuck(code, inp, ctx=ctx, log_max_rows=24)
        t = time.perf_counter()
        for _ in range(10):
            pkg.prove_brainfuck(code, inp, ctx=ctx, log_max_rows=24)
        print(f"{name:18s} LOG_MAX_ROWS=24 reuse_preprocessed={reuse}: {(time.perf_counter() - t) * 100:.2f} ms per proof (VM + tables + proof + JSON)")
    pkg.lib().bfhip_ctx_reuse_preprocessed(ctx._h, 0)
ctx.close()
