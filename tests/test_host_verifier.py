"""CPU: the product's own verifier (bfhip_verify_brainfuck, host only) agrees with the oracle's verifier on accepted and on tampered
proofs. Proofs come from the oracle prover here (no GPU in this suite); the GPU suite feeds it HIP proofs."""
import re

import pytest


@pytest.fixture(scope="module")
def proof(oracle):
    js, _, _ = oracle.prove("+++>,<[>+.<-]", b"\x01", log_max_rows=12)
    return js


def test_accepts_valid_proof(pkg, oracle, proof):
    assert pkg.verify_brainfuck(proof, 12) == (True, "")
    assert oracle.verify(proof, 12)[0]


def test_accepts_other_programs(pkg, oracle):
    for code, inp in [("++[-]+.", b""), (",[.,]", b"ab\x00"), ("+", b"")]:
        js, _, _ = oracle.prove(code, inp, log_max_rows=12)
        assert pkg.verify_brainfuck(js, 12)[0]


def _tamper(js, key, which):
    start = js.index(key)
    m = list(re.finditer(rb"\d+", js[start:]))[which]
    return js[: start + m.start()] + str(int(m.group()) + 1).encode() + js[start + m.end():]


@pytest.mark.parametrize("key,which", [
    (b'"commitments"', 3), (b'"sampled_values"', 0), (b'"sampled_values"', 40), (b'"queried_values"', 2), (b'"proof_of_work"', 0),
    (b'"fri_witness"', 1), (b'"hash_witness"', 5), (b'"column_witness"', 0), (b'"coeffs"', 0), (b'"claimed_sum"', 0), (b'"log_size"', 0),
])
def test_rejects_tampering_like_the_oracle(pkg, oracle, proof, key, which):
    bad = _tamper(proof, key, which)
    ok, reason = pkg.verify_brainfuck(bad, 12)
    ook, oreason = oracle.verify(bad, 12)
    assert not ok and not ook
    assert reason.split(":")[0] == oreason.split(":")[0]      # same VerificationError class


def test_rejects_garbage(pkg):
    assert not pkg.verify_brainfuck(b"{}", 12)[0]
    assert not pkg.verify_brainfuck(b"not json", 12)[0]
    assert not pkg.verify_brainfuck(b"", 12)[0]


def test_rejects_wrong_log_max_rows(pkg, proof):
    assert not pkg.verify_brainfuck(proof, 13)[0]
