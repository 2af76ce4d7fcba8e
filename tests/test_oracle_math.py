"""CPU: algebraic pins for the oracle's field / circle-FFT / Merkle / channel restatement (parity with stwo is unpinned — SURVEY F5 —
so these are identities and public KATs, not reference vectors)."""
import ctypes
import hashlib

import numpy as np
import pytest

from conftest import P, splitmix_column


def m31(oracle, op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    out = np.zeros_like(a)
    bp = None if b is None else np.ascontiguousarray(b, dtype=np.uint32).ctypes.data_as(ctypes.c_void_p)
    oracle.L.orc_m31_op(op, a.ctypes.data_as(ctypes.c_void_p), bp, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(a.size))
    return out


def qm31(oracle, op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    out = np.zeros_like(a)
    bp = None if b is None else np.ascontiguousarray(b, dtype=np.uint32).ctypes.data_as(ctypes.c_void_p)
    oracle.L.orc_qm31_op(op, a.ctypes.data_as(ctypes.c_void_p), bp, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(a.size // 4))
    return out


def test_m31_against_python_ints(oracle):
    a, b = splitmix_column(1, 4096), splitmix_column(2, 4096)
    ai, bi = a.astype(object), b.astype(object)
    assert m31(oracle, 0, a, b).tolist() == [(x + y) % P for x, y in zip(ai, bi)]
    assert m31(oracle, 1, a, b).tolist() == [(x - y) % P for x, y in zip(ai, bi)]
    assert m31(oracle, 2, a, b).tolist() == [(x * y) % P for x, y in zip(ai, bi)]
    a[a == 0] = 1
    assert m31(oracle, 3, a).tolist() == [pow(int(x), P - 2, P) for x in a]
    # edge values
    e = np.array([0, 1, P - 1, P - 1, 2], dtype=np.uint32); f = np.array([0, P - 1, P - 1, 1, 1 << 30], dtype=np.uint32)
    assert m31(oracle, 2, e, f).tolist() == [(int(x) * int(y)) % P for x, y in zip(e, f)]


def test_inverse_of_two_is_reference_constant(oracle):
    # BaseField::from(2).inverse() as used symbolically by the reference (machine.rs:427)
    assert m31(oracle, 3, np.array([2], dtype=np.uint32))[0] == 1073741824


def test_qm31_product_known_answer(oracle):
    """(1 + 2i + (3 + 4i)u) * (4 + 5i + (6 + 7i)u) = -71 + 93i + (-16 + 50i)u with u^2 = 2 + i — the worked product of upstream stwo's
    qm31 unit test (core/fields/qm31.rs test_ops: qm31!(P - 71, 93, P - 16, 50)); checkable by hand."""
    a = np.array([1, 2, 3, 4], dtype=np.uint32); b = np.array([4, 5, 6, 7], dtype=np.uint32)
    assert qm31(oracle, 2, a, b).tolist() == [P - 71, 93, P - 16, 50]
    assert qm31(oracle, 0, a, b).tolist() == [5, 7, 9, 11]
    assert qm31(oracle, 1, a, b).tolist() == [P - 3] * 4


def test_qm31_field_axioms(oracle):
    a = splitmix_column(3, 4 * 512); b = splitmix_column(4, 4 * 512); c = splitmix_column(5, 4 * 512)
    mul = lambda x, y: qm31(oracle, 2, x, y)
    add = lambda x, y: qm31(oracle, 0, x, y)
    assert np.array_equal(mul(a, b), mul(b, a))
    assert np.array_equal(mul(mul(a, b), c), mul(a, mul(b, c)))
    assert np.array_equal(mul(a, add(b, c)), add(mul(a, b), mul(a, c)))
    one = np.tile(np.array([1, 0, 0, 0], dtype=np.uint32), 512)
    assert np.array_equal(mul(a, qm31(oracle, 3, a)), one)
    # u^2 = 2 + i  (QM31 = CM31[u]/(u^2 - 2 - i)), i^2 = -1
    u = np.array([0, 0, 1, 0], dtype=np.uint32); i = np.array([0, 1, 0, 0], dtype=np.uint32)
    assert mul(u, u).tolist() == [2, 1, 0, 0]
    assert mul(i, i).tolist() == [P - 1, 0, 0, 0]


@pytest.mark.parametrize("log", [3, 4, 7, 10, 13])
def test_fft_roundtrip_and_point_evaluation(oracle, log):
    cols = np.stack([splitmix_column(10 + log, 1 << log), splitmix_column(20 + log, 1 << log)])
    coeffs = oracle.interpolate(cols, log)
    assert np.array_equal(oracle.evaluate(coeffs, log, log), cols)            # evaluate . interpolate = id on the same domain
    lde = oracle.evaluate(coeffs, log, log + 1)
    # eval_at_point(interpolate(f), domain.at(i)) == f[bit_reverse(i)] on the trace domain and on the LDE domain
    for domain_log, values in [(log, cols), (log + 1, lde)]:
        for idx in [0, 1, 5, (1 << domain_log) - 1]:
            xy = (ctypes.c_uint32 * 2)()
            oracle.L.orc_domain_point(domain_log, idx, xy)
            pt = (ctypes.c_uint32 * 8)(xy[0], 0, 0, 0, xy[1], 0, 0, 0)
            out = (ctypes.c_uint32 * 4)()
            oracle.L.orc_eval_at_point(coeffs[0].ctypes.data_as(ctypes.c_void_p), log, pt, out)
            br = int(format(idx, f"0{domain_log}b")[::-1], 2)
            assert list(out) == [int(values[0][br]), 0, 0, 0]


def test_fft_is_linear(oracle):
    log = 9
    a, b = splitmix_column(31, 1 << log), splitmix_column(32, 1 << log)
    s = ((a.astype(np.uint64) + b) % P).astype(np.uint32)
    ca, cb, cs = (oracle.interpolate(x[None, :], log)[0] for x in (a, b, s))
    assert np.array_equal(((ca.astype(np.uint64) + cb) % P).astype(np.uint32), cs)


def test_replicated_column_has_sparse_coefficients(oracle):
    """A column whose values are broadcast 16x (memory/table.rs:95-104) interpolates to coefficients supported on indices = 0 mod 16,
    and its LDE is again 16x replicated — the structure the HIP path exploits."""
    log = 10
    rows = splitmix_column(77, 1 << (log - 4))
    coeffs = oracle.interpolate(np.repeat(rows, 16)[None, :], log)[0]
    assert not coeffs.reshape(-1, 16)[:, 1:].any()
    lde = oracle.evaluate(coeffs[None, :], log, log + 1)[0].reshape(-1, 16)
    assert np.all(lde == lde[:, :1])


def test_python_compress_reproduces_hashlib():
    """Pins conftest.py_blake2s_compress (the F function both node-hash conventions are built from) to hashlib's Blake2s-256 on one-
    and two-block messages: h0 = IV ^ 0x01010020, byte counter t, final flag."""
    import struct
    from conftest import _B2S_IV, py_blake2s_compress
    for msg in [b"", b"abc", bytes(range(64)), bytes(range(100)), bytes(200)]:
        h = list(_B2S_IV); h[0] ^= 0x01010020
        blocks = [msg[i:i + 64] for i in range(0, max(len(msg), 1), 64)] or [b""]
        t = 0
        for k, b in enumerate(blocks):
            last = k == len(blocks) - 1
            t += len(b)
            h = py_blake2s_compress(h, list(struct.unpack("<16I", b.ljust(64, b"\0"))), t0=t, f0=0xFFFFFFFF if last else 0)
        assert struct.pack("<8I", *h) == hashlib.blake2s(msg).digest()


@pytest.mark.parametrize("n_vals", [0, 1, 4, 15, 16, 17, 33])
def test_hash_node_matches_python_restatement(oracle, conv, n_vals):
    """Blake2sMerkleHasher::hash_node of the oracle under the current convention == the independent Python restatement, with and without
    children, over the chunk boundaries of the zero-padding rule (rem = 15 - ((len + 15) % 16))."""
    from conftest import py_hash_node
    vals = splitmix_column(900 + n_vals, max(n_vals, 1))[:n_vals]
    l, r = hashlib.sha256(b"l").digest(), hashlib.sha256(b"r").digest()
    assert oracle.hash_node(l, r, vals) == py_hash_node(conv[0], l, r, vals)
    if n_vals:
        assert oracle.hash_node(None, None, vals) == py_hash_node(conv[0], None, None, vals)
    if conv[0] == 0 and n_vals == 0:
        assert oracle.hash_node(None, None, vals) == bytes(32)      # nothing absorbed: the zero state


def test_merkle_root_matches_python_restatement(oracle, conv):
    """Mixed-degree tree: node = hash_node(left, right, values of the columns of that layer's size), commit order kept inside a size."""
    from conftest import py_hash_node
    big = [splitmix_column(40 + k, 8) for k in range(3)]     # log 3
    small = [splitmix_column(50, 2)]                           # log 1
    cols = [big[0], small[0], big[1], big[2]]                  # commit order is preserved inside a size class
    H = lambda l, r, v: py_hash_node(conv[0], l, r, v)
    layer = [H(None, None, [c[i] for c in big]) for i in range(8)]
    layer = [H(layer[2 * i], layer[2 * i + 1], []) for i in range(4)]
    layer = [H(layer[2 * i], layer[2 * i + 1], [small[0][i]]) for i in range(2)]
    root = H(layer[0], layer[1], [])
    ptrs = (ctypes.c_void_p * 4)(*[c.ctypes.data for c in cols])
    logs = (ctypes.c_uint32 * 4)(3, 1, 3, 3)
    out = (ctypes.c_ubyte * 32)()
    assert oracle.L.orc_merkle_commit(ptrs, logs, ctypes.c_size_t(4), out, None) == 0
    assert bytes(out) == root


def test_channel_primitives(oracle, conv):
    ch = ctypes.c_void_p(oracle.L.orc_channel_new())
    d = (ctypes.c_ubyte * 32)()
    oracle.L.orc_channel_digest(ch, d)
    assert bytes(d) == bytes(32)                                # Blake2sChannel::default()
    root = bytes(range(32))
    oracle.L.orc_channel_mix_root(ch, root)
    oracle.L.orc_channel_digest(ch, d)
    assert bytes(d) == hashlib.blake2s(bytes(32) + root).digest()   # mix_root = H(digest || root)
    before = bytes(d)
    felts = np.array([1, 2, 3, 4, 5, 6, 7, 8], dtype=np.uint32)
    oracle.L.orc_channel_mix_felts(ch, felts.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(2))
    oracle.L.orc_channel_digest(ch, d)
    assert bytes(d) == hashlib.blake2s(before + felts.tobytes()).digest()
    before = bytes(d)
    out = (ctypes.c_uint32 * 4)()
    oracle.L.orc_channel_draw_felt(ch, out)
    words = np.frombuffer(hashlib.blake2s(before + bytes(32)).digest(), dtype=np.uint32)   # counter 0, zero padded
    if np.all(words < 2 * P):
        assert list(out) == [int(w % P) for w in words[:4]]
    nonce = oracle.L.orc_channel_grind(ch, 5)
    oracle.L.orc_channel_digest(ch, d)
    before = bytes(d)
    oracle.L.orc_channel_mix_u64(ch, ctypes.c_uint64(nonce))
    assert oracle.L.orc_channel_trailing_zeros(ch) >= 5
    oracle.L.orc_channel_digest(ch, d)
    import struct
    from conftest import py_blake2s_compress
    if conv[1] == 0:    # mix_u64 = raw compression of [lo, hi, 0 x 14] on the digest words
        want = struct.pack("<8I", *py_blake2s_compress(list(struct.unpack("<8I", before)), [nonce & 0xFFFFFFFF, nonce >> 32] + [0] * 14))
    else:               # Blake2s-256(digest || LE64(n) zero padded to 32 bytes)
        want = hashlib.blake2s(before + struct.pack("<Q", nonce) + bytes(24)).digest()
    assert bytes(d) == want
    oracle.L.orc_channel_free(ch)
