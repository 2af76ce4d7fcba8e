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


# ---- the circle-FFT conventions restated from their DEFINITION in pure Python --------------------------------------
# Independent of the oracle's and the product's code: circle group law on the generator (2, 1268011823), CanonicCoset(n) = odds(n) with
# circle_domain() = half_odds(n - 1) followed by its conjugates, evaluations stored bit-reversed, monomial basis y^b0 x^b1 pi(x)^b2 ...
# with pi(x) = 2 x^2 - 1. O(n^2) evaluation by the definition, small sizes only.
def _cadd(p, q):
    return ((p[0] * q[0] - p[1] * q[1]) % P, (p[0] * q[1] + p[1] * q[0]) % P)


def _cpow(g, e):
    r, b = (1, 0), g
    while e:
        if e & 1:
            r = _cadd(r, b)
        b = _cadd(b, b)
        e >>= 1
    return r


_GEN = (2, 1268011823)        # M31_CIRCLE_GEN, order 2^31


def _domain_point(log, i):
    """CanonicCoset(log).circle_domain().at(i): half_coset = Coset(initial = subgroup_gen(log + 1), step = subgroup_gen(log - 1), log - 1);
    at(i) = half_coset.at(i) for i < 2^(log-1), else the conjugate of half_coset.at(i - 2^(log-1)). subgroup_gen(k) = G^(2^(31-k))."""
    half = 1 << (log - 1)
    j = i if i < half else i - half
    idx = ((1 << (31 - (log + 1))) + j * (1 << (31 - (log - 1)))) % (1 << 31)
    x, y = _cpow(_GEN, idx)
    return (x, y) if i < half else (x, (P - y) % P)


def _basis(log, j, pt):
    """j-th basis function of a circle polynomial of 2^log coefficients at pt: bit 0 of j selects y, bit k >= 1 selects pi^(k-1)(x)."""
    x, y = pt
    r = y if j & 1 else 1
    f = x
    for k in range(1, log):
        if (j >> k) & 1:
            r = r * f % P
        f = (2 * f * f - 1) % P
    return r


@pytest.mark.parametrize("log", [3, 4, 6])
def test_fft_conventions_match_the_definition(oracle, log):
    n = 1 << log
    assert (_cpow(_GEN, 1 << 30), _cpow(_GEN, 1 << 31)) == ((P - 1, 0), (1, 0))       # the generator has order 2^31
    for i in (0, 1, n // 2, n - 1):                                                   # the oracle's domain points are the defined ones
        xy = (ctypes.c_uint32 * 2)()
        oracle.L.orc_domain_point(log, i, xy)
        assert (xy[0], xy[1]) == _domain_point(log, i)
    coeffs = splitmix_column(4242 + log, n)
    br = lambda i: int(format(i, f"0{log}b")[::-1], 2)
    want = np.zeros(n, dtype=np.uint32)
    for i in range(n):                                                                # stored index bit_reverse(i) holds f(domain.at(i))
        pt = _domain_point(log, i)
        want[br(i)] = sum(int(coeffs[j]) * _basis(log, j, pt) for j in range(n)) % P
    assert np.array_equal(oracle.evaluate(coeffs[None, :], log, log)[0], want)        # evaluate = the definition
    assert np.array_equal(oracle.interpolate(want[None, :], log)[0], coeffs)          # interpolate = its inverse
    # LDE: the same polynomial (coefficients zero-extended) on the canonic domain of twice the size
    lde_want = np.zeros(2 * n, dtype=np.uint32)
    brl = lambda i: int(format(i, f"0{log + 1}b")[::-1], 2)
    for i in range(2 * n):
        pt = _domain_point(log + 1, i)
        lde_want[brl(i)] = sum(int(coeffs[j]) * _basis(log + 1, j, pt) for j in range(n)) % P
    assert np.array_equal(oracle.evaluate(coeffs[None, :], log, log + 1)[0], lde_want)


def _qmul_m(q, m):
    return [x * m % P for x in q]


def _qadd(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def _qmul(a, b):
    """QM31 product from the definition: (a0 + a1 i) + (a2 + a3 i) u, i^2 = -1, u^2 = 2 + i."""
    def cm(x, y):
        return ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)
    a0, a1, b0, b1 = (a[0], a[1]), (a[2], a[3]), (b[0], b[1]), (b[2], b[3])
    t = cm(a1, b1)
    r = ((2 * t[0] - t[1]) % P, (t[0] + 2 * t[1]) % P)                 # (2 + i) * a1 * b1
    lo = cm(a0, b0); hi0 = cm(a0, b1); hi1 = cm(a1, b0)
    return [(lo[0] + r[0]) % P, (lo[1] + r[1]) % P, (hi0[0] + hi1[0]) % P, (hi0[1] + hi1[1]) % P]


@pytest.mark.parametrize("log", [2, 3, 5])
def test_fold_line_matches_the_definition(oracle, log):
    """FriOps::fold_line from its definition: for p(x) = p0(pi(x)) + x p1(pi(x)) given by its evaluations on LineDomain(half_odds(log))
    (bit-reversed), the fold with alpha is 2 (p0 + alpha p1) on the doubled domain — no 1/2 normalisation."""
    n = 1 << log
    br = lambda i, l: int(format(i, f"0{l}b")[::-1], 2) if l else 0
    def line_x(l, i):       # x-coordinate of Coset::half_odds(l).at(i): initial subgroup_gen(l + 2), step subgroup_gen(l)
        return _cpow(_GEN, ((1 << (31 - (l + 2))) + i * (1 << (31 - l))) % (1 << 31))[0]
    def basis(l, j, x):     # line basis: bit k of j selects pi^k(x)
        r, f = 1, x
        for k in range(l):
            if (j >> k) & 1:
                r = r * f % P
            f = (2 * f * f - 1) % P
        return r
    c = [int(v) for v in splitmix_column(777 + log, n)]
    alpha = [int(v) for v in splitmix_column(778 + log, 4)]
    src = [np.zeros(n, dtype=np.uint32) for _ in range(4)]
    for i in range(n):
        src[0][br(i, log)] = sum(c[j] * basis(log, j, line_x(log, i)) for j in range(n)) % P     # base-field polynomial embedded in QM31
    want = [np.zeros(n // 2, dtype=np.uint32) for _ in range(4)]
    for i in range(n // 2):
        y = line_x(log - 1, i)
        p0 = sum(c[2 * j] * basis(log - 1, j, y) for j in range(n // 2)) % P
        p1 = sum(c[2 * j + 1] * basis(log - 1, j, y) for j in range(n // 2)) % P
        val = _qadd([2 * p0 % P, 0, 0, 0], _qmul_m(alpha, 2 * p1 % P))
        for k in range(4):
            want[k][br(i, log - 1)] = val[k]
    got = [np.zeros(n // 2, dtype=np.uint32) for _ in range(4)]
    sp = (ctypes.c_void_p * 4)(*[a.ctypes.data for a in src]); dp = (ctypes.c_void_p * 4)(*[a.ctypes.data for a in got])
    assert oracle.L.orc_fold_line(sp, log, (ctypes.c_uint32 * 4)(*alpha), dp) == 0
    for k in range(4):
        assert np.array_equal(got[k], want[k]), k


@pytest.mark.parametrize("log", [3, 4])
def test_fold_circle_into_line_matches_the_definition(oracle, log):
    """FriOps::fold_circle_into_line from its definition: f(x, y) = g0(x) + y g1(x) on CanonicCoset(log).circle_domain() (bit-reversed) folds
    to dst * alpha^2 + 2 (g0 + alpha g1) on LineDomain(half_odds(log - 1))."""
    n = 1 << log
    br = lambda i, l: int(format(i, f"0{l}b")[::-1], 2) if l else 0
    c = [int(v) for v in splitmix_column(881 + log, n)]
    alpha = [int(v) for v in splitmix_column(882 + log, 4)]
    dst0 = [[int(v) for v in splitmix_column(883 + k, n // 2)] for k in range(4)]
    src = [np.zeros(n, dtype=np.uint32) for _ in range(4)]
    for i in range(n):
        src[0][br(i, log)] = sum(c[j] * _basis(log, j, _domain_point(log, i)) for j in range(n)) % P
    a2 = _qmul(alpha, alpha)
    want = [np.zeros(n // 2, dtype=np.uint32) for _ in range(4)]
    def xbasis(l, j, x):    # basis of the line polynomial in x: bit k of j selects pi^k(x)
        r, f = 1, x
        for k in range(l):
            if (j >> k) & 1:
                r = r * f % P
            f = (2 * f * f - 1) % P
        return r
    for i in range(n // 2):
        x = _domain_point(log, i)[0]          # the first half of the circle domain is the half-coset: its x-coordinates are the line domain
        g0 = sum(c[2 * j] * xbasis(log - 1, j, x) for j in range(n // 2)) % P
        g1 = sum(c[2 * j + 1] * xbasis(log - 1, j, x) for j in range(n // 2)) % P
        slot = br(i, log - 1)
        val = _qadd(_qmul([dst0[k][slot] for k in range(4)], a2), _qadd([2 * g0 % P, 0, 0, 0], _qmul_m(alpha, 2 * g1 % P)))
        for k in range(4):
            want[k][slot] = val[k]
    dst = [np.array(dst0[k], dtype=np.uint32) for k in range(4)]
    sp = (ctypes.c_void_p * 4)(*[a.ctypes.data for a in src]); dp = (ctypes.c_void_p * 4)(*[a.ctypes.data for a in dst])
    assert oracle.L.orc_fold_circle_into_line(dp, sp, log, (ctypes.c_uint32 * 4)(*alpha)) == 0
    for k in range(4):
        assert np.array_equal(dst[k], want[k]), k


# ---- oracle/simd_bound.cpp: the host-vector-unit bound of bench.py's cpu_baseline ---------------------------------------------------------
def test_simd_bound_compression_is_rfc7693_and_vector_paths_agree(oracle):
    """The 16-lane Blake2s loop whose rate bounds the reference's CPU path computes the RFC 7693 compression (its scalar instance against the
    hashlib-pinned Python restatement), the AVX-512 and AVX2 instances agree lane for lane, and the packed M31 butterflies equal plain modular
    arithmetic (skipped per instruction set on hosts that lack it)."""
    import ctypes
    from conftest import py_blake2s_compress, _B2S_IV
    L = oracle.L
    L.orc_simd_bound_blake_scalar.restype = ctypes.c_uint32
    L.orc_simd_bound_blake_scalar.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int]
    seed, iters = 0xC0FFEE, 5
    M = 0xFFFFFFFF
    h = [_B2S_IV[i] ^ ((seed + i) & M) for i in range(8)]
    m = [(seed * 2654435761 + i) & M for i in range(16)]
    for _ in range(iters):
        h = py_blake2s_compress(h, m, t0=64, f0=0)
        m[0] ^= h[0]; m[5] ^= h[3]; m[10] ^= h[5]; m[15] ^= h[7]
    assert [L.orc_simd_bound_blake_scalar(iters, seed, w) for w in range(8)] == h
    assert L.orc_simd_bound_selfcheck(0) in (0, -1) and L.orc_simd_bound_selfcheck(1) in (0, -1)


def test_simd_bound_runs_and_reports_rates(oracle):
    import ctypes
    L = oracle.L
    L.orc_simd_bound.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
    out = (ctypes.c_double * 4)()
    L.orc_simd_bound(2, 0.2, out)
    assert out[3] == 2 and (out[2] == 0 or (out[0] > 1e6 and out[1] > 1e7)), list(out)
