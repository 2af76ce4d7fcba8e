import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_package
pkg = load_package()
code = open(os.path.join(ROOT, 'tests/golden/programs/fib19.bf')).read()
t0 = time.time(); ctx = pkg.Context(0, max_log_domain=26); print('ctx', time.time()-t0, flush=True)
t0 = time.time(); tr = pkg.Trace(ctx, code); print('trace', time.time()-t0, tr.log_sizes, tr.n_steps, tr.cells, flush=True)
for i in range(3):
    t0 = time.time(); proof, tm = tr.prove(24); dt = time.time()-t0
    print(i, 'prove %.3fs' % dt, 'cells/s %.3e' % (tr.cells/dt), {k: round(v*1e3,1) for k,v in tm.items()}, len(proof), flush=True)
import hashlib; print(hashlib.sha256(proof).hexdigest())

os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
open(os.path.join(ROOT, 'gpurun_out/fib19_proof.json'), 'wb').write(proof)
