"""CPU: the Poseidon252 oracle (oracle/poseidon252.py) against the public Hades known-answer vector and its own sponge rules."""
import importlib.util
import os

from conftest import ROOT

spec = importlib.util.spec_from_file_location("poseidon252_oracle", os.path.join(ROOT, "oracle", "poseidon252.py"))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def test_hades_public_known_answer():
    ref.self_test()


def _header_words(name):
    import re
    txt = open(os.path.join(ROOT, "stwo-brainfuck_amd", "csrc", "poseidon_constants.h")).read()
    body = txt[txt.index(" " + name + "["):]
    body = body[body.index("=") + 1:body.index(";")]
    return [int(x.rstrip("u"), 16) for x in re.findall(r"0x[0-9a-f]{8}u", body)]


def test_generated_constants_header_matches_oracle():
    """stwo-brainfuck_amd/csrc/poseidon_constants.h, host set: the round constants, R and R^2 in Montgomery form with R = 2^256 (8 words)."""
    def val(ws): return sum(w << (32 * i) for i, w in enumerate(ws))
    flat = [k for r in ref.ARK for k in r]
    words = _header_words("POSEIDON_ARK")
    assert len(words) == 273 * 8
    for j, k in enumerate(flat):
        assert val(words[8 * j:8 * j + 8]) == k * 2**256 % ref.P
    assert val(_header_words("POSEIDON_R1")) == 2**256 % ref.P and val(_header_words("POSEIDON_R2")) == 2**512 % ref.P
    assert val(_header_words("POSEIDON_P")) == ref.P


def test_partial_round_recurrence():
    """The identity poseidon_dev.h's partial rounds rest on, in plain big integers: 83 rounds of (s0, s1, s2) -> MDS * (s0 + k0, s1 + k1, (s2 + k2)^3)
    equal the second-order sequence T^_r = 2 T^_(r-1) + 4 T^_(r-2) + y_r, s2' = 2 T^_(r-1) + Lambda_r - 2 y_r started from (s0 + s1) / 2,
    (s0 - s1) / 4 and ended by s0 = T^_86 + 2 T^_85 + A, s1 = T^_86 - 2 T^_85 + B (constants as tools/gen_poseidon_constants.py derives them)."""
    import random
    P = ref.P
    k = [tuple(r) for r in ref.ARK]
    r0, r1 = 4, 86
    sig = lambda r: (k[r][0] + k[r][1]) % P
    E = {r: 0 if r < r0 else 4 * k[r][0] % P if r == r0 else (4 * sig(r - 1) + 4 * k[r][0]) % P for r in range(r0 - 2, r1 + 1)}
    delta = {r0 - 1: 0, r0 - 2: 0}
    for r in range(r0, r1 + 1):
        delta[r] = (2 * delta[r - 1] + 4 * delta[r - 2] - (E[r - 1] + 2 * E[r - 2])) % P
    lam = {r: (E[r - 1] + sig(r) - 2 * delta[r - 1]) % P for r in range(r0, r1 + 1)}
    inv2 = pow(2, -1, P)
    A = (-delta[r1] + E[r1] * inv2 - 2 * delta[r1 - 1] + E[r1 - 1] + sig(r1)) % P
    B = (-delta[r1] + E[r1] * inv2 + 2 * delta[r1 - 1] - E[r1 - 1] - sig(r1)) % P
    rng = random.Random(83)
    for _ in range(6):
        a, b, c = (rng.randrange(P) for _ in range(3))
        x, y_, z = a, b, c
        for r in range(r0, r1 + 1):
            cube = pow(z + k[r][2], 3, P)
            x, y_, z = (3 * (x + k[r][0]) + (y_ + k[r][1]) + cube) % P, ((x + k[r][0]) - (y_ + k[r][1]) + cube) % P, ((x + k[r][0]) + (y_ + k[r][1]) - 2 * cube) % P
        T1, T2, cc = (a + b) * inv2 % P, (a - b) * inv2 * inv2 % P, c
        for r in range(r0, r1 + 1):
            cube = pow(cc + k[r][2], 3, P)
            T1, T2, cc = (2 * T1 + 4 * T2 + cube) % P, T1, (2 * T1 + lam[r] - 2 * cube) % P
        assert ((T1 + 2 * T2 + A) % P, (T1 - 2 * T2 + B) % P, cc) == (x, y_, z)


def _device_table():
    table = _header_words("POSEIDON_DEV_ROUNDS")
    assert len(table) == 92 * 54
    return table


def test_generated_device_round_table_is_the_same_permutation():
    """Device set (R = 2^261, 9 limbs of 29 bits; full rounds K0 K1 K2 L0 L1 L2, partial rounds K2 and Lc, row 91 = S8 . . A B .): a big-int walk
    through the table with exactly the kernel's formulas (poseidon_dev.h hades_round / hades_partial*) gives the oracle's Hades on Montgomery
    forms, and every table entry has the documented shape."""
    P, R = ref.P, 2**261
    def val(ls): return sum(l << (29 * i) for i, l in enumerate(ls))
    table = _device_table()
    assert val(_header_words("POSEIDON_DEV_R1")) == R % P and val(_header_words("POSEIDON_DEV_R2")) == R * R % P
    cube = lambda x: x * x * x * pow(R, -2, P) % P                                                  # two Montgomery products
    inv2 = pow(2, -1, P)
    last = table[54 * 91:54 * 92]
    assert val(last[0:9]) == 8 * P and all(2**29 - 1 <= l for l in last[0:8]) and last[8] >= 2**22 - 1
    assert val(last[27:36]) % P == (val(last[27:36]) - 2 * P) % P and all(2**29 - 1 <= l for l in last[27:35]) and all(3 * 2**29 - 3 <= l for l in last[36:44])
    for st in ([0, 0, 0], [1, 2, 3], [P - 1, 5, P - 2]):
        s = [x * R % P for x in st]
        T1 = T2 = None
        for r in range(91):
            row = table[54 * r:54 * r + 54]
            K = [val(row[9 * i:9 * i + 9]) for i in range(3)]
            L = [val(row[27 + 9 * i:36 + 9 * i]) for i in range(3)]
            full = r < 4 or r >= 87
            assert all(l < 2**29 for i in range(3) for l in row[9 * i:9 * i + 8])                      # K: normalised
            if full:
                assert all(2**29 - 1 <= l for l in row[27:35]) and all(2**30 - 2 <= l for l in row[36:44]) and all(3 * 2**29 - 3 <= l for l in row[45:53])
                c0, c1, c2 = cube(s[0] + K[0]), cube(s[1] + K[1]), cube(s[2] + K[2])
                s = [(3 * c0 + c1 + c2 + L[0]) % P, (c0 + c2 + L[1] - c1) % P, (c0 + c1 + L[2] - 2 * c2) % P]
                continue
            assert K[0] == 0 and K[1] == 0 and L[0] == 0 and L[1] == 0 and all(3 * 2**29 - 3 <= l for l in row[45:53])
            if r == 4:
                T1, T2 = (s[0] + s[1]) * inv2 % P, (s[0] + val(last[0:9]) - s[1]) * inv2 * inv2 % P
            y = cube(s[2] + K[2])
            T1, T2, s[2] = (2 * T1 + 4 * T2 + y) % P, T1, (2 * T1 + L[2] - 2 * y) % P
            if r == 86:
                s[0], s[1] = (T1 + 2 * T2 + val(last[27:36])) % P, (T1 + val(last[36:45]) - 2 * T2) % P
        assert [x * pow(R, -1, P) % P for x in s] == ref.hades(list(st))


def test_sponge_padding_rules():
    # poseidon_hash_many: [] -> hades([1,0,0])[0]; [a] -> hades([a,1,0])[0]; [a,b] -> hades(hades([a,b,0]) + [1,0,0])[0]
    assert ref.poseidon_hash_many([]) == ref.hades([1, 0, 0])[0]
    assert ref.poseidon_hash_many([5]) == ref.hades([5, 1, 0])[0]
    s = ref.hades([5, 7, 0]); s[0] = (s[0] + 1) % ref.P
    assert ref.poseidon_hash_many([5, 7]) == ref.hades(s)[0]


def test_hash_node_packing():
    # 8 values pack into one felt as w = w * 2^31 + v; a 9th value opens a second zero-padded block
    vals = list(range(1, 9))
    w = 0
    for v in vals:
        w = w * 2**31 + v
    assert ref.hash_node(None, vals) == ref.poseidon_hash_many([w])
    assert ref.hash_node(None, vals + [9]) == ref.poseidon_hash_many([w, 9 * 2**(31 * 7)])
    assert ref.hash_node((3, 4), []) == ref.poseidon_hash_many([3, 4])


# ---- the C++ oracle's own felt252 / Hades / sponge / channel (oracle/poseidon252.h, blake2s.h) against the big-integer Python above ----
import ctypes

import numpy as np
import pytest


def _limbs(x):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def _int(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def test_cpp_hades_matches_python(_oracle):
    import random
    rng = random.Random(5)
    for state in [[0, 0, 0], [1, 2, 3], [ref.P - 1, ref.P - 2, 5]] + [[rng.randrange(ref.P) for _ in range(3)] for _ in range(5)]:
        inp = (ctypes.c_uint64 * 12)(*[w for x in state for w in _limbs(x)])
        out = (ctypes.c_uint64 * 12)()
        _oracle.L.orc_hades(inp, out)
        assert [_int(out[4 * k: 4 * k + 4]) for k in range(3)] == ref.hades(state)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 7])
def test_cpp_hash_many_matches_python(_oracle, n):
    vals = [(0x1234567 * (k + 1)) ** 5 % ref.P for k in range(n)]
    inp = (ctypes.c_uint64 * max(4 * n, 4))(*[w for x in vals for w in _limbs(x)])
    out = (ctypes.c_uint64 * 4)()
    _oracle.L.orc_poseidon_hash_many(inp, ctypes.c_size_t(n), out)
    assert _int(out) == ref.poseidon_hash_many(vals)


@pytest.mark.parametrize("n_vals", [0, 1, 7, 8, 9, 17])
def test_cpp_hash_node_matches_python(_oracle, n_vals):
    from conftest import splitmix_column
    _oracle.set_conventions(0, 0, 0, 1)
    try:
        vals = splitmix_column(77 + n_vals, max(n_vals, 1))[:n_vals]
        l, r = 12345678901234567890 ** 3 % ref.P, 98765432109876543210 ** 3 % ref.P
        lb, rb = l.to_bytes(32, "little"), r.to_bytes(32, "little")
        assert int.from_bytes(_oracle.hash_node(lb, rb, vals), "little") == ref.hash_node((l, r), [int(v) for v in vals])
        if n_vals:
            assert int.from_bytes(_oracle.hash_node(None, None, vals), "little") == ref.hash_node(None, [int(v) for v in vals])
    finally:
        _oracle.set_conventions(0, 0, 0, 0)


def test_cpp_poseidon_channel_matches_python_restatement(_oracle):
    """Poseidon252Channel (stwo core/channel/poseidon252.rs, recalled — parity unpinned) restated with Python integers."""
    M31 = (1 << 31) - 1
    _oracle.set_conventions(0, 0, 0, 1)
    try:
        L = _oracle.L
        ch = ctypes.c_void_p(L.orc_channel_new())
        d = (ctypes.c_ubyte * 32)()
        digest, n_sent = 0, 0

        def check():
            L.orc_channel_digest(ch, d)
            assert int.from_bytes(bytes(d), "little") == digest
        check()
        root = 0x123456789abcdef ** 4 % ref.P
        L.orc_channel_mix_root(ch, root.to_bytes(32, "little")); digest, n_sent = ref.hades([digest, root, 2])[0], 0; check()          # poseidon_hash(digest, root)
        L.orc_channel_mix_u64(ch, ctypes.c_uint64(0xdeadbeefcafe)); digest, n_sent = ref.hades([digest, 0xdeadbeefcafe, 2])[0], 0; check()
        felts = np.array([1, 2, 3, 4, 5, 6, 7, 8, M31 - 1, 0, 11, 12], dtype=np.uint32)                                              # 3 secure felts -> 2 chunks
        L.orc_channel_mix_felts(ch, felts.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(3))
        words = [digest]
        for chunk in (felts[:8], felts[8:]):
            w = 0
            for y in chunk:
                w = w * 2**31 + int(y)
            words.append(w)
        digest, n_sent = ref.poseidon_hash_many(words), 0; check()
        out = (ctypes.c_uint32 * 4)()
        L.orc_channel_draw_felt(ch, out)
        drawn = ref.hades([digest, n_sent, 2])[0]; n_sent += 1
        want = [((drawn >> (31 * i)) & M31) % M31 for i in range(4)]
        assert list(out) == want
        tz = L.orc_channel_trailing_zeros(ch)
        be = digest.to_bytes(32, "big")
        v = int.from_bytes(be[:16], "little")
        assert tz == (128 if v == 0 else (v & -v).bit_length() - 1)
        L.orc_channel_free(ch)
    finally:
        _oracle.set_conventions(0, 0, 0, 0)


def test_limb_level_model_of_the_device_hades():
    """poseidon_dev.h restated limb by limb (9 limbs of 29 bits, u32 limbs, u64 column accumulators) on the committed round table: every
    intermediate stays inside the width the kernel gives it — column sums < 2^64, linear-layer limbs in [0, 2^32), nothing negative after the
    subtraction of (q - 1) p, normalised limbs < 2^29 with the top limb < 2^22, values < 3 p (< 4 p for the two T of the partial rounds at
    their start), the T update of the partial rounds never above 2^32 - 1 although it uses all but 7 of the 2^32 values — and the permutation
    equals the oracle's, for zero, small, near-p, all-limbs-saturated and random states entering in weak form (up to + 5 p)."""
    import random
    P, R, M29, U32 = ref.P, 2**261, (1 << 29) - 1, 0xffffffff
    C6 = 17 << 18
    table = _device_table()

    def limbs9(x): return [(x >> (29 * i)) & M29 for i in range(8)] + [x >> 232]
    def val(l): return sum(v << (29 * i) for i, v in enumerate(l))

    def f9_mul(a, b):
        r, m, acc = [0] * 9, [0] * 9, 0
        for K in range(17):
            for i in range(0 if K < 9 else K - 8, (K if K < 9 else 8) + 1):
                acc += a[i] * b[K - i]
            if 6 <= K < 15: acc += m[K - 6] * C6
            if K >= 8: acc += m[K - 8] << 19
            assert acc < 2**64
            if K < 9:
                m[K] = (-(acc & U32)) & M29; acc += m[K]
                assert acc % (1 << 29) == 0
            else:
                r[K - 9] = acc & M29
            acc >>= 29
        r[8] = acc
        assert acc < 2**24
        return r

    def cube(a):
        assert all(0 <= x < (1 << 30) for x in a) and val(a) < 2**256
        return f9_mul(f9_mul(a, a), a)

    def normalise(o, top=1 << 25):
        o = list(o)
        assert all(0 <= x <= U32 for x in o)
        for i in range(8):
            o[i + 1] += o[i] >> 29; o[i] &= M29
            assert o[i + 1] <= U32
        assert o[8] < top
        return o

    def reduce(o):
        assert all(0 <= x <= U32 for x in o)
        q = ((o[8] + (o[7] >> 29)) & U32) >> 19
        k = max(q, 1) - 1
        assert k < (1 << 24)                                     # the kernel multiplies k * C6 with the 24-bit multiplier
        o = list(o)
        o[0] -= k; o[6] -= k * C6; o[8] -= k << 19
        assert all(x >= 0 for x in o)
        for i in range(8):
            o[i + 1] += o[i] >> 29; o[i] &= M29
        assert o[8] < (1 << 22) and val(o) < 3 * P
        return o

    def half(x):                                                # f9_half: normalised in, lazily normalised out
        assert all(v < (1 << 29) for v in x[:8])
        odd = x[0] & 1
        r = [(x[i] >> 1) | ((x[i + 1] & 1) << 28) for i in range(8)] + [x[8] >> 1]
        if odd:
            r[0] += 1; r[6] += 17 << 17; r[8] += 1 << 18
        assert val(r) * 2 % P == val(x) % P
        return r

    def partial(T1, T2, c, t):
        assert all(v < (1 << 29) for v in T1[:8] + T2[:8] + c[:8]) and val(T1) < 4 * P and val(T2) < 4 * P and val(c) < 3 * P
        y = cube([a + k for a, k in zip(c, t[18:27])])
        cn = [2 * a + l - 2 * b for a, b, l in zip(T1, y, t[45:54])]
        return t_update(T1, T2, y), T1, reduce(cn)

    def t_update(T1, T2, y):
        Tn = [2 * a + 4 * b + d for a, b, d in zip(T1, T2, y)]
        assert all(0 <= v <= 7 * M29 for v in Tn[:8]) and Tn[8] < (1 << 25)
        q = Tn[8] >> 19
        k = max(q, 1) - 1
        assert k <= 26
        # u32 arithmetic as the kernel does it: limb 6 and limb 8 may pass below zero before their carry and bias arrive
        Tn[0] = (Tn[0] + (1 << 29) - k) & U32; Tn[6] = (Tn[6] - k * C6) & U32; Tn[8] = (Tn[8] - (k << 19) - 1) & U32
        want = 2 * val(T1) + 4 * val(T2) + val(y) - k * P
        assert want >= 0
        for i in range(8):
            add = (Tn[i] >> 29) + (M29 if i < 7 else 0)
            assert i == 5 or i == 7 or Tn[i + 1] + add <= U32     # the only wraps allowed are those that repair limbs 6 and 8
            Tn[i + 1] = (Tn[i + 1] + add) & U32; Tn[i] &= M29
        assert val(Tn) == want and Tn[8] < (1 << 22) and want < 3 * (1 << 251)
        return Tn

    def hades9(s):
        last = table[54 * 91:54 * 92]
        for r in range(91):
            t = table[54 * r:54 * r + 54]
            if r < 4 or r >= 87:
                K = [t[9 * i:9 * i + 9] for i in range(3)]
                L = [t[27 + 9 * i:36 + 9 * i] for i in range(3)]
                c0, c1, c2 = (cube([a + k for a, k in zip(s[i], K[i])]) for i in range(3))
                s = [reduce([3 * a + b + c + l for a, b, c, l in zip(c0, c1, c2, L[0])]),
                     reduce([a + c + l - b for a, b, c, l in zip(c0, c1, c2, L[1])]),
                     reduce([a + b + l - 2 * c for a, b, c, l in zip(c0, c1, c2, L[2])])]
                continue
            if r == 4:                                          # hades_partial_begin
                dif = [a + k - b for a, b, k in zip(s[0], s[1], last[0:9])]
                assert all(v >= 0 for v in dif)
                T1 = normalise(half(normalise([a + b for a, b in zip(s[0], s[1])])))
                T2 = normalise(half(normalise(half(normalise(dif)))))
                assert val(T1) * 2 % P == (val(s[0]) + val(s[1])) % P and val(T2) * 4 % P == (val(s[0]) - val(s[1])) % P
                c = s[2]
            T1, T2, c = partial(T1, T2, c, t)
            if r == 86:                                         # hades_partial_end
                a = [x + 2 * y + k for x, y, k in zip(T1, T2, last[27:36])]
                b = [x + k - 2 * y for x, y, k in zip(T1, T2, last[36:45])]
                s = [reduce(a), reduce(b), c]
        return s

    # the T update at its limits: every limb of T1, T2 and y saturated (8 * 2^29 - 7 on limb 0, 8 * 2^29 - 8 + carry 7 above), top limbs at the
    # largest values the contracts allow, and the all-zero case (q = 0: the bias alone must not leave limb 8 below zero)
    for top in (0, 1, (1 << 21) - 1):
        full9 = [M29] * 8 + [top]
        t_update(full9, full9, [M29] * 8 + [min(top, (1 << 20) - 1)])
    t_update([0] * 9, [0] * 9, [0] * 9)
    t_update([0] * 9, [0] * 9, [1] + [0] * 8)

    rng = random.Random(29)
    sat = val([M29] * 8 + [0])                                   # every limb below the top one saturated
    states = [[0, 0, 0], [1, 2, 3], [P - 1, P - 1, P - 1], [P - 1, 0, 1], [(1 << 232) - 1, 1 << 232, (1 << 29) - 1], [sat, sat, sat], [sat, 0, sat]] + \
             [[rng.randrange(P) for _ in range(3)] for _ in range(12)]
    for st in states:
        want = ref.hades(list(st))
        for extra in (0, 5):                                  # the sponge hands over states below 6 p
            out = hades9([limbs9(x * R % P + extra * P) for x in st])
            assert [val(o) * pow(R, -1, P) % P for o in out] == want
