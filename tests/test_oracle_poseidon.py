"""CPU: the Poseidon252 oracle (oracle/poseidon252.py) against the public Hades known-answer vector and its own sponge rules."""
import importlib.util
import os

from conftest import ROOT

spec = importlib.util.spec_from_file_location("poseidon252_oracle", os.path.join(ROOT, "oracle", "poseidon252.py"))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def test_hades_public_known_answer():
    ref.self_test()


def test_generated_constants_header_matches_oracle():
    """stwo-brainfuck_amd/csrc/poseidon_constants.h holds the same round constants (Montgomery form, x * 2^256 mod p)."""
    import re
    txt = open(os.path.join(ROOT, "stwo-brainfuck_amd", "csrc", "poseidon_constants.h")).read()
    body = txt[txt.index("POSEIDON_ARK"):]
    rows = re.findall(r"\{((?:0x[0-9a-f]{8}u,? ?){8})\}", body)
    assert len(rows) == 273
    flat = [k for r in ref.ARK for k in r]
    for row, k in zip(rows, flat):
        limbs = [int(x.rstrip("u"), 16) for x in re.findall(r"0x[0-9a-f]{8}u", row)]
        assert sum(l << (32 * i) for i, l in enumerate(limbs)) == k * 2**256 % ref.P


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
