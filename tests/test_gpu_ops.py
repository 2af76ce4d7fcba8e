"""-m gpu parity of the single backend operations exported by include/bfhip.h against the oracle / exact integer arithmetic."""
import ctypes
import hashlib

import numpy as np
import pytest

from conftest import splitmix_column, P

pytestmark = pytest.mark.gpu


def test_twiddles_match_oracle(ctx, oracle):
    tw, itw, root_log = ctx.twiddles()
    k = 14                                   # tail of the tree = twiddle buffer of Coset::half_odds(k)
    got = ctx.download(tw + 4 * ((1 << root_log) - (1 << k)), 1 << k)
    goti = ctx.download(itw + 4 * ((1 << root_log) - (1 << k)), 1 << k)
    want = np.zeros(1 << k, dtype=np.uint32); wanti = np.zeros(1 << k, dtype=np.uint32)
    assert oracle.L.orc_twiddles(k, want.ctypes.data_as(ctypes.c_void_p), wanti.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.array_equal(got, want) and np.array_equal(goti, wanti)


def test_bit_reverse_and_gather(ctx):
    log = 15
    col = splitmix_column(5, 1 << log)
    p, q = ctx.upload(col), ctx.malloc(4 << log)
    ctx.bit_reverse(p, q, log)
    got = ctx.download(q, 1 << log)
    idx = np.array([int(format(i, f"0{log}b")[::-1], 2) for i in range(1 << log)])
    assert np.array_equal(got[idx], col)
    pos = np.array([0, 1, 77, (1 << log) - 1, 12345], dtype=np.uint64)
    assert np.array_equal(ctx.gather(p, pos), col[pos.astype(np.int64)])
    ctx.free(p); ctx.free(q)


@pytest.mark.parametrize("n", [1, 7, 8, 1000, 1 << 16])
def test_batch_inverse_m31(ctx, n):
    col = splitmix_column(9, n); col[col == 0] = 1
    p, q = ctx.upload(col), ctx.malloc(4 * n)
    ctx.batch_inverse_m31(p, q, n)
    got = ctx.download(q, n)
    assert np.all((got.astype(object) * col.astype(object)) % P == 1)
    ctx.free(p); ctx.free(q)


@pytest.mark.parametrize("n", [1, 3, 4, 1001, 1 << 15])
def test_batch_inverse_qm31(ctx, oracle, n):
    cols = [splitmix_column(20 + k, n) for k in range(4)]
    cols[0][cols[0] == 0] = 1
    flat = np.ascontiguousarray(np.stack(cols, axis=1).reshape(-1))           # AoS for the oracle's qm31 op
    want = np.zeros_like(flat)
    oracle.L.orc_qm31_op(3, flat.ctypes.data_as(ctypes.c_void_p), None, want.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n))
    src = [ctx.upload(c) for c in cols]
    ctx.batch_inverse_qm31(src, src, n)                                        # in place
    got = np.stack([ctx.download(p, n) for p in src], axis=1).reshape(-1)
    assert np.array_equal(got, want)
    for p in src:
        ctx.free(p)


def test_broadcast16_then_full_transform_equals_replicated_transform(ctx):
    log = 12
    rows = splitmix_column(66, 1 << (log - 4))
    pr, pf = ctx.upload(rows), ctx.malloc(4 << log)
    ctx.broadcast16(pr, pf, rows.size)
    assert np.array_equal(ctx.download(pf, 1 << log), np.repeat(rows, 16))
    ctx.interpolate([pf], [pf], log)                      # full circle iFFT of the broadcast column
    ctx.interpolate([pr], [pr], log, replicated=True)     # line-mode iFFT of the row-granular column
    full = ctx.download(pf, 1 << log)
    assert np.array_equal(full[::16], ctx.download(pr, 1 << (log - 4))) and not full.reshape(-1, 16)[:, 1:].any()
    ctx.free(pr); ctx.free(pf)


def test_accumulate(ctx):
    a, b = splitmix_column(1, 5000), splitmix_column(2, 5000)
    pa, pb = ctx.upload(a), ctx.upload(b)
    ctx.accumulate(pa, pb, 5000)
    assert np.array_equal(ctx.download(pa, 5000), ((a.astype(np.uint64) + b) % P).astype(np.uint32))
    ctx.free(pa); ctx.free(pb)


@pytest.mark.parametrize("log,replicated", [(4, False), (9, False), (13, False), (17, False), (5, True), (12, True), (18, True)])
def test_eval_at_point(ctx, oracle, log, replicated):
    n = 1 << (log - 4 if replicated else log)
    coeffs = splitmix_column(100 + log, n)
    full = coeffs
    if replicated:
        full = np.zeros(1 << log, dtype=np.uint32); full[::16] = coeffs      # coefficients of a 16x-replicated column
    point = splitmix_column(7, 8)   # any QM31 pair works for the fold; it need not lie on the circle
    out = (ctypes.c_uint32 * 4)()
    oracle.L.orc_eval_at_point(full.ctypes.data_as(ctypes.c_void_p), log, (ctypes.c_uint32 * 8)(*point.tolist()), out)
    p = ctx.upload(coeffs)
    assert ctx.eval_at_point(p, log, point, replicated) == list(out)
    ctx.free(p)


def test_merkle_layers_match_oracle(ctx, oracle):
    """Mixed-degree tree built layer by layer with bfhip_merkle_commit_layer == oracle MerkleProver::commit (all layers)."""
    logs = [10, 10, 10, 8, 8, 5]
    cols = [splitmix_column(200 + i, 1 << l) for i, l in enumerate(logs)]
    ptrs_h = (ctypes.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
    total = sum(32 << l for l in range(11))
    want_layers = np.zeros(total, dtype=np.uint8); root = (ctypes.c_ubyte * 32)()
    assert oracle.L.orc_merkle_commit(ptrs_h, (ctypes.c_uint32 * len(logs))(*logs), ctypes.c_size_t(len(logs)), root, want_layers.ctypes.data_as(ctypes.c_void_p)) == 0
    dev = [ctx.upload(c) for c in cols]
    prev, off = 0, 0
    for log in range(10, -1, -1):
        layer_cols = [d for d, l in zip(dev, logs) if l == log]
        out = ctx.malloc(32 << log)
        ctx.merkle_commit_layer(log, prev, layer_cols, out)
        got = ctx.download(out, 8 << log).view(np.uint8)
        assert np.array_equal(got, want_layers[off: off + (32 << log)]), f"layer {log}"
        off += 32 << log
        prev = out
    assert bytes(ctx.download(prev, 8).view(np.uint8)) == bytes(root)


def test_merkle_leaf_matches_python_restatement_with_replicated_column(ctx, conv):
    """Leaf layer of the GPU kernel against the independent Python restatement of the node hash (conftest.py_hash_node), both conventions."""
    from conftest import py_hash_node
    log = 6
    full = splitmix_column(31, 1 << log); rows = splitmix_column(32, 1 << (log - 4))
    pf, pr = ctx.upload(full), ctx.upload(rows)
    out = ctx.malloc(32 << log)
    ctx.merkle_commit_layer(log, 0, [pf, pr], out, col_shifts=[0, 4])
    got = ctx.download(out, 8 << log).view(np.uint8).reshape(-1, 32)
    for i in range(1 << log):
        assert bytes(got[i]) == py_hash_node(conv[0], None, None, [full[i], rows[i >> 4]])


@pytest.mark.parametrize("ncols", [0, 1, 5, 16, 17, 40])
def test_merkle_inner_nodes_match_python_restatement(ctx, conv, ncols):
    """Nodes with children and 0..40 columns (1..4 compressions, the zero-padding boundaries at 16 and 17 columns) against the Python
    restatement, both conventions."""
    from conftest import py_hash_node
    log = 3
    prev_bytes = np.frombuffer(b"".join(hashlib.sha256(bytes([k])).digest() for k in range(2 << log)), dtype=np.uint8)
    cols = [splitmix_column(700 + k, 1 << log) for k in range(ncols)]
    dprev = ctx.upload(prev_bytes)
    dcols = [ctx.upload(c) for c in cols]
    out = ctx.malloc(32 << log)
    ctx.merkle_commit_layer(log, dprev, dcols, out)
    got = ctx.download(out, 8 << log).view(np.uint8).reshape(-1, 32)
    pb = prev_bytes.tobytes()
    for i in range(1 << log):
        l, r = pb[64 * i: 64 * i + 32], pb[64 * i + 32: 64 * i + 64]
        assert bytes(got[i]) == py_hash_node(conv[0], l, r, [c[i] for c in cols])


@pytest.mark.parametrize("log", [1, 2, 5, 11, 16])
def test_fold_line(ctx, oracle, log):
    src = [splitmix_column(300 + k, 1 << log) for k in range(4)]
    alpha = splitmix_column(3, 4)
    want = [np.zeros(1 << (log - 1), dtype=np.uint32) for _ in range(4)]
    sp = (ctypes.c_void_p * 4)(*[s.ctypes.data for s in src]); dp = (ctypes.c_void_p * 4)(*[w.ctypes.data for w in want])
    assert oracle.L.orc_fold_line(sp, log, (ctypes.c_uint32 * 4)(*alpha.tolist()), dp) == 0
    ds = [ctx.upload(s) for s in src]; dd = [ctx.malloc(4 << (log - 1)) for _ in range(4)]
    ctx.fold_line(ds, dd, log, alpha)
    for k in range(4):
        assert np.array_equal(ctx.download(dd[k], 1 << (log - 1)), want[k])


@pytest.mark.parametrize("log", [3, 6, 12, 17])
def test_fold_circle_into_line(ctx, oracle, log):
    src = [splitmix_column(400 + k, 1 << log) for k in range(4)]
    dst = [splitmix_column(500 + k, 1 << (log - 1)) for k in range(4)]
    alpha = splitmix_column(4, 4)
    want = [d.copy() for d in dst]
    sp = (ctypes.c_void_p * 4)(*[s.ctypes.data for s in src]); dp = (ctypes.c_void_p * 4)(*[w.ctypes.data for w in want])
    assert oracle.L.orc_fold_circle_into_line(dp, sp, log, (ctypes.c_uint32 * 4)(*alpha.tolist())) == 0
    ds = [ctx.upload(s) for s in src]; dd = [ctx.upload(d) for d in dst]
    ctx.fold_circle_into_line(dd, ds, log, alpha)
    for k in range(4):
        assert np.array_equal(ctx.download(dd[k], 1 << (log - 1)), want[k])


@pytest.mark.parametrize("pow_bits", [0, 5, 12, 20])
def test_grind_finds_smallest_nonce(ctx, oracle, pow_bits):
    digest = hashlib.blake2s(b"bfhip grind test %d" % pow_bits).digest()
    oracle.L.orc_grind_digest.restype = ctypes.c_uint64
    want = oracle.L.orc_grind_digest(digest, pow_bits)
    assert ctx.grind(digest, pow_bits) == want
