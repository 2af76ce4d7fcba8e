"""Times the prover-input preparation (VM + 13 table builders + upload) and the end-to-end prove for fib19.bf."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package()
code = open(os.path.join(ROOT, "tests/golden/programs/fib19.bf")).read()
ctx = pkg.Context(0, max_log_domain=26)
for i in range(4):
    t0 = time.time(); tr = pkg.Trace(ctx, code); t1 = time.time(); proof, ph = tr.prove(24); t2 = time.time(); proof, ph2 = tr.prove(24); t3 = time.time(); tr.close()
    print('second prove on same trace %.1f ms' % (1e3*(t3-t2)))
    print({k: round(v*1e3,1) for k,v in ph.items()})
    print(f"trace_create {1e3*(t1-t0):.1f} ms, prove {1e3*(t2-t1):.1f} ms, end-to-end {1e3*(t2-t0):.1f} ms -> {tr.cells/(t2-t0):.3e} cells/s (PCIe inclusive)", flush=True)
ctx.close()

ctx = pkg.Context(0, max_log_domain=26)
for i in range(4):
    t0 = time.time(); proof = pkg.prove_brainfuck(code, b"", ctx=ctx, log_max_rows=24); t1 = time.time()
    print(f"one-call prove_brainfuck (VM + tables + upload overlapped with the preprocessed phase): {1e3*(t1-t0):.1f} ms -> {403753616/(t1-t0):.3e} cells/s", flush=True)
ctx.close()
