"""CPU: the oracle's full prove -> verify path on the reference's end-to-end programs (brainfuck_air/mod.rs:804-858), tamper
rejection, AIR satisfiability (memory/component.rs:163-209, plus_component.rs:145-190) and negative AIR cases."""
import json
import os
import re

import pytest

pytestmark = pytest.mark.with_poseidon     # also under the Poseidon252MerkleChannel variant (BASELINE config 5)

import numpy as np

from conftest import ROOT, P, splitmix_column

LMR = 12   # LOG_MAX_ROWS for the CPU suite: the four e2e programs fit (largest component log 10..12); keeps a proof under a second

E2E = [
    ("+++>,<[>+.<-]", b"\x01"),
    ("++++++++++[>+++++++>++++++++++>+++>+<<<<-]>++.>+.+++++++..+++.>++.<<+++++++++++++++.>.+++.------.--------.>+.>.", b""),
    ("+++><[>+<-]", b""),
    ("++[-]+.", b""),
]


_PROOFS = {}   # convention set -> proofs (the oracle fixture is per test: it follows the convention parametrisation)


@pytest.fixture
def proofs(oracle, conv):
    if conv not in _PROOFS:
        out = []
        for code, inp in E2E:
            ls, _ = oracle.log_sizes(code, inp)
            lmr = max(LMR, max(ls))
            js, tr, _ = oracle.prove(code, inp, log_max_rows=lmr)
            out.append((js, lmr))
        _PROOFS[conv] = out
    return _PROOFS[conv]


def test_prove_verify_roundtrip(oracle, proofs):
    for js, lmr in proofs:
        ok, err = oracle.verify(js, lmr)
        assert ok, err


def test_proof_is_deterministic(oracle):
    a, _, _ = oracle.prove(*E2E[0], log_max_rows=LMR)
    b, _, _ = oracle.prove(*E2E[0], log_max_rows=LMR)
    assert a == b


def test_proof_json_shape(proofs, conv):
    p = json.loads(proofs[0][0])
    assert list(p.keys()) == ["claim", "interaction_claim", "proof"]                       # BrainfuckProof (mod.rs:71-76)
    assert list(p["claim"].keys())[:4] == ["memory", "instruction", "program", "processor"]   # BrainfuckClaim order (mod.rs:85-99)
    assert p["claim"]["memory"] == {"log_size": p["claim"]["memory"]["log_size"], "_marker": None}
    if conv[3] == 1:      # Poseidon252MerkleHasher::Hash = FieldElement252: "0x" + minimal lowercase hex (starknet-ff's human-readable serde form)
        assert len(p["proof"]["commitments"]) == 4 and all(re.fullmatch(r"0x(0|[1-9a-f][0-9a-f]{0,62})", c) and int(c, 16) < 2**251 + 17 * 2**192 + 1 for c in p["proof"]["commitments"])
    else:
        assert len(p["proof"]["commitments"]) == 4 and all(len(c) == 32 for c in p["proof"]["commitments"])
    assert [len(t) for t in p["proof"]["sampled_values"]][1:] == [128, 60, 4]               # 128 main, 60 interaction, 4 composition columns
    # the last logUp column of each component is sampled at two points (offsets 0 and -1), all other columns at one
    n_two = sum(1 for c in p["proof"]["sampled_values"][2] if len(c) == 2)
    assert n_two == 13 * 4
    assert p["proof"]["fri_proof"]["last_layer_poly"]["log_size"] == 0 and len(p["proof"]["fri_proof"]["last_layer_poly"]["coeffs"]) == 1


def test_lookup_sums_cancel(proofs):
    # lookup_sum_valid (mod.rs:207-227): the 13 claimed sums add up to zero in QM31
    P = (1 << 31) - 1
    p = json.loads(proofs[0][0])
    tot = [0, 0, 0, 0]
    for v in p["interaction_claim"].values():
        (a, b), (c, d) = v["claimed_sum"]
        tot = [(x + y) % P for x, y in zip(tot, (a, b, c, d))]
    assert tot == [0, 0, 0, 0]


def _tamper_number(js: bytes, key: bytes, which: int) -> bytes:
    """Adds 1 to the `which`-th integer literal after the first occurrence of `key`."""
    start = js.index(key)
    ms = list(re.finditer(rb"\d+", js[start:]))
    m = ms[which]
    val = int(m.group()) + 1
    return js[: start + m.start()] + str(val).encode() + js[start + m.end():]


@pytest.mark.parametrize("key,which", [
    (b'"commitments"', 3), (b'"sampled_values"', 0), (b'"sampled_values"', 40), (b'"queried_values"', 2), (b'"proof_of_work"', 0),
    (b'"fri_witness"', 1), (b'"hash_witness"', 5), (b'"column_witness"', 0), (b'"coeffs"', 0), (b'"claimed_sum"', 0), (b'"log_size"', 0),
])
def test_tampered_proof_is_rejected(oracle, proofs, key, which):
    js, lmr = proofs[0]
    bad = _tamper_number(js, key, which)
    assert bad != js
    ok, err = oracle.verify(bad, lmr)
    assert not ok, f"tampering {key!r}[{which}] was accepted"


def test_truncated_proof_is_rejected(oracle, proofs):
    js, lmr = proofs[0]
    ok, _ = oracle.verify(js[: len(js) // 2], lmr)
    assert not ok


def test_wrong_log_max_rows_is_rejected(oracle, proofs):
    js, lmr = proofs[0]
    ok, _ = oracle.verify(js, lmr + 1)   # different preprocessed tree
    assert not ok


# ---- AIR satisfiability (stwo assert_constraints analogue) -------------------------------------------------------------------------------
@pytest.mark.parametrize("component", range(13))
def test_air_satisfied_on_real_trace(oracle, component):
    # memory/component.rs:163-209 uses "+>,<[>+.<-]" with input 1 and dummy lookup elements
    rc, row, c = oracle.assert_constraints("+>,<[>+.<-]", b"\x01", component)
    assert rc == 0, f"constraint {c} fails at row {row}"


def test_air_satisfied_with_non_trivial_lookup_elements(oracle):
    elems = [5, 1, 2, 3, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83]
    for component in (0, 3, 10):
        rc, row, c = oracle.assert_constraints("+++>,<[>+.<-]", b"\x01", component, elems=elems)
        assert rc == 0, f"component {component}: constraint {c} fails at row {row}"


@pytest.mark.parametrize("component,col,row,val,constraint", [
    (0, 3, 1, 2, 4),       # memory: d = 2 violates d(d-1)                      (memory/component.rs negative tests :211-609)
    (0, 0, 0, 1, 0),       # memory: first clk != 0 violates is_first * clk
    (1, 3, 0, 2, 1),       # instruction: d = 2 violates d(d-1)
    (3, 8, 2, 9, 6),       # processor: next_clk - clk - 1 != 0
    (10, 2, 0, 44, 0),     # plus: ci is not '+' (and not 0)
])
def test_air_rejects_corrupted_trace(oracle, component, col, row, val, constraint):
    elems = [5, 1, 2, 3, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83]   # avoid zero logUp denominators
    rc, bad_row, bad_c = oracle.assert_constraints("+>,<[>+.<-]", b"\x01", component, elems=elems, corrupt=(col, row, val))
    assert rc == 1 and bad_c == constraint and bad_row // 16 == row


def test_per_component_checkers_are_consistent(oracle):
    """The per-component oracle entry points (checkers of bfhip_logup_generate / bfhip_eval_constraints): the 13 claimed sums
    cancel for shared lookup elements, replicated logUp columns are 16-lane broadcasts, and constraint evaluation is additive in
    the accumulator and linear in the random coefficients."""
    code, inp = "+++>,<[>+.<-]", b"\x01"
    elems = splitmix_column(31, 24).tolist()
    total = np.zeros(4, dtype=object)
    for comp in range(13):
        rows = np.ascontiguousarray(oracle.table(code, inp, comp).T)
        cols, claimed = oracle.logup_generate(comp, rows, elems)
        for k in range(cols.shape[0] - 4):
            assert np.array_equal(np.repeat(cols[k][::16], 16), cols[k])
        assert cols[-4:, 1].tolist() == claimed                      # prefix_sum.at(1) is the last coset element (finalize_last)
        total = (total + np.array(claimed, dtype=object)) % P
    assert not total.any()

    comp = 0
    rows = np.ascontiguousarray(oracle.table(code, inp, comp).T)
    log = int(np.log2(rows.shape[1])) + 4
    n = 1 << (log + 1)
    inter, claimed = oracle.logup_generate(comp, rows, elems)
    lde = lambda c: oracle.evaluate(oracle.interpolate(c, log), log, log + 1)
    one_hot = np.zeros((1, 1 << log), dtype=np.uint32); one_hot[0, 0] = 1
    args = (comp, log, lde(one_hot)[0], lde(np.repeat(rows, 16, axis=1)), lde(inter), elems, claimed)
    c1, c2 = splitmix_column(1, 48), splitmix_column(2, 48)
    zero = np.zeros((4, n), dtype=np.uint32)
    add = lambda a, b: ((a.astype(np.uint64) + b) % P).astype(np.uint32)
    r1, r2 = oracle.eval_constraints(*args, c1, zero), oracle.eval_constraints(*args, c2, zero)
    assert np.array_equal(oracle.eval_constraints(*args, add(c1, c2), zero), add(r1, r2))
    assert np.array_equal(oracle.eval_constraints(*args, c2, r1), add(r1, r2))
