"""Throughput of the Poseidon252 Merkle leaf kernel: 2^log leaves x C columns (BASELINE config 5 hasher)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package, splitmix_column
pkg = load_package()
ctx = pkg.Context(0, max_log_domain=12)
log, C = 22, 8
cols = [ctx.upload(splitmix_column(k, 1 << log)) for k in range(C)]
out = ctx.malloc(32 << log)
for rep in range(3):
    ctx.sync(); t0 = time.time(); ctx.merkle_commit_layer_poseidon252(log, 0, cols, out); ctx.sync(); dt = time.time() - t0
perms = (1 << log) * 1          # 8 columns -> 1 block -> poseidon_hash_many([w]) = 1 permutation
print(f"2^{log} leaves x {C} cols: {dt*1e3:.2f} ms -> {perms/dt/1e6:.1f} M Hades permutations/s, {(1<<log)*C*4/dt/1e9:.2f} GB/s of column data")
blake = ctx.malloc(32 << log)
ctx.sync(); t0 = time.time(); ctx.merkle_commit_layer(log, 0, cols, blake); ctx.sync(); dtb = time.time() - t0
print(f"same layer with Blake2s: {dtb*1e3:.3f} ms ({dt/dtb:.0f}x faster)")
