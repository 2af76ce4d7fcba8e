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
