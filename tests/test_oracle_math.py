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


def test_merkle_root_matches_hashlib(oracle):
    """Mixed-degree tree: node = blake2s(left || right || LE u32 of the columns of that layer's size)."""
    big = [splitmix_column(40 + k, 8) for k in range(3)]     # log 3
    small = [splitmix_column(50, 2)]                           # log 1
    cols = [big[0], small[0], big[1], big[2]]                  # commit order is preserved inside a size class
    layer = [hashlib.blake2s(b"".join(int(c[i]).to_bytes(4, "little") for c in big)).digest() for i in range(8)]
    layer = [hashlib.blake2s(layer[2 * i] + layer[2 * i + 1]).digest() for i in range(4)]
    layer = [hashlib.blake2s(layer[2 * i] + layer[2 * i + 1] + int(small[0][i]).to_bytes(4, "little")).digest() for i in range(2)]
    root = hashlib.blake2s(layer[0] + layer[1]).digest()
    ptrs = (ctypes.c_void_p * 4)(*[c.ctypes.data for c in cols])
    logs = (ctypes.c_uint32 * 4)(3, 1, 3, 3)
    out = (ctypes.c_ubyte * 32)()
    assert oracle.L.orc_merkle_commit(ptrs, logs, ctypes.c_size_t(4), out, None) == 0
    assert bytes(out) == root


def test_channel_primitives(oracle):
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
    oracle.L.orc_channel_mix_u64(ch, ctypes.c_uint64(nonce))
    assert oracle.L.orc_channel_trailing_zeros(ch) >= 5
    oracle.L.orc_channel_free(ch)
