import sys, time, os
sys.path.insert(0, "/root/repo/tests")
from conftest import load_package
pkg = load_package()
ctx = pkg.Context(0, max_log_domain=26)
code = open("/root/repo/tests/golden/programs/fib19.bf").read()
tr = pkg.Trace(ctx, code, b"")
for want in (True, False, True, False):
    for _ in range(2): tr.prove(24, want_json=want)
    t = time.perf_counter()
    for _ in range(10): tr.prove(24, want_json=want)
    print("want_json", want, round((time.perf_counter() - t) * 100, 3), "ms/proof")
