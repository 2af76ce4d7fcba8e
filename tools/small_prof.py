import os, sys, time
sys.path.insert(0, "/root/repo/tests")
from conftest import load_package
pkg = load_package()
ctx = pkg.Context(0, max_log_domain=26)
code = open("/root/repo/tests/golden/programs/hello_kakarot.bf").read()
pkg.lib().bfhip_ctx_reuse_preprocessed(ctx._h, 1)
for _ in range(4):
    pkg.prove_brainfuck(code, b"", ctx=ctx, log_max_rows=24)
_, ph = pkg.prove_brainfuck(code, b"", ctx=ctx, log_max_rows=24, with_timings=True)
print({k: round(v*1e3,2) for k,v in ph.items()})
ctx.close()
