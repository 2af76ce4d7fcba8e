"""CPU: the product's own verifier (bfhip_verify_brainfuck, host only) agrees with the oracle's verifier on accepted and on tampered
proofs. Proofs come from the oracle prover here (no GPU in this suite); the GPU suite feeds it HIP proofs."""
import re

import pytest

pytestmark = pytest.mark.with_poseidon     # also under the Poseidon252MerkleChannel variant (BASELINE config 5)


_PROOF = {}


@pytest.fixture
def proof(oracle, conv):
    if conv not in _PROOF:
        _PROOF[conv] = oracle.prove("+++>,<[>+.<-]", b"\x01", log_max_rows=12)[0]
    return _PROOF[conv]


def test_proof_verifies_only_under_its_own_conventions(pkg, oracle, proof, conv):
    """A proof is bound to the byte-level conventions it was produced under (include/bfhip.h bfhip_conventions)."""
    from conftest import CONVENTIONS
    for other in CONVENTIONS.values():
        ok, _ = pkg.verify_brainfuck(proof, 12, conventions=other)
        assert ok == (tuple(other) == tuple(conv))


def test_verifier_rejects_non_canonical_json(pkg, proof):
    """serde_json (the reference's reader, bin/brainfuck_prover.rs:146-151) rejects these; so must the product's reader."""
    import re as _re
    assert pkg.verify_brainfuck(proof + b" ", 12)[0]                        # trailing whitespace is fine
    assert not pkg.verify_brainfuck(proof + b"x", 12)[0]                    # trailing bytes
    assert not pkg.verify_brainfuck(proof + proof, 12)[0]
    m = _re.search(rb'"log_size":(\d+)', proof)
    wrapped = str(int(m.group(1)) + (1 << 64)).encode()                      # wraps to the same value modulo 2^64
    assert not pkg.verify_brainfuck(proof[: m.start(1)] + wrapped + proof[m.end(1):], 12)[0]
    assert not pkg.verify_brainfuck(proof[: m.start(1)] + b"0" + m.group(1) + proof[m.end(1):], 12)[0]   # leading zero
    assert not pkg.verify_brainfuck(proof.replace(b"null", b"nuII", 1), 12)[0]


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


def test_verifiers_agree_on_mutated_proofs(pkg, oracle, proof, conv):
    """Differential check (a fixed slice of tools/fuzz_verifier.py): structural and textual mutations of a valid proof get the same verdict
    from the product's verifier and the oracle's — here always a rejection. The campaign found the oracle's JSON reader wrapping 2^64 + v to v
    and ignoring bytes after the proof object (round 2); both readers now take exactly what serde_json takes."""
    import json, os, random, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from fuzz_verifier import mutate, text_mutations
    rng = random.Random(7)
    tree = json.loads(proof)
    cases = [(json.dumps(m, separators=(",", ":")).encode(), what) for m, what in filter(None, (mutate(tree, rng) for _ in range(120)))]
    cases += text_mutations(proof, rng)
    assert len(cases) > 80
    for js, what in cases:
        a = pkg.verify_brainfuck(js, 12, conv)[0]
        b = oracle.verify(js, 12)[0]
        assert a == b, what
        assert not a, f"both verifiers accept the proof after: {what}"


def test_last_layer_log_size_must_match_its_coefficients(pkg, oracle, proof, conv):
    """last_layer_poly.log_size is not bound by the transcript; the reference's LinePoly::eval_at_point folds 2^log_size coefficients and panics on
    any other count. Both verifiers refuse a log_size that does not match (found by tools/fuzz_verifier.py: 0 -> 1 was accepted by both)."""
    bad = proof.replace(b'"log_size":0}}}}', b'"log_size":1}}}}')
    assert bad != proof
    assert not pkg.verify_brainfuck(bad, 12, conv)[0]
    assert not oracle.verify(bad, 12)[0]


@pytest.mark.single_conv
def test_bfprove_verify_try_all_names_the_convention_set(_oracle, tmp_path):
    """tools/bfprove.py verify --try-all (host only): the one-command pin for whoever holds a proof written by the Rust reference
    (`brainfuck_prover prove --output`, bin/brainfuck_prover.rs:127-131) — it must name exactly the switch set a proof was made under,
    for Blake2s and Poseidon252 proofs alike, and list the first failing check per set for a proof nothing accepts."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = [sys.executable, os.path.join(root, "tools", "bfprove.py"), "verify"]
    try:
        for conv in ((0, 0, 0, 0), (1, 0, 1, 0), (0, 0, 1, 1)):
            _oracle.set_conventions(*conv)
            proof = _oracle.prove("++>,<[>+.<-]", b"\x02", log_max_rows=10)[0]
            f = tmp_path / ("p%d%d%d%d.json" % conv)
            f.write_bytes(proof)
            r = subprocess.run(tool + [str(f), "--log-max-rows", "10", "--try-all"], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stdout + r.stderr
            hits = [l for l in r.stdout.splitlines() if l.startswith("Proof verified under")]
            assert len(hits) == 1 and ("--conventions %d,%d,%d,%d " % conv) in hits[0], r.stdout
            # the explicit form of the same switch set
            r = subprocess.run(tool + [str(f), "--log-max-rows", "10", "--conventions", "%d,%d,%d,%d" % conv], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0 and "Proof verified" in r.stdout
        f.write_bytes(proof.replace(b'"proof_of_work":', b'"proof_of_work":1', 1))
        r = subprocess.run(tool + [str(f), "--log-max-rows", "10", "--try-all"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 1 and r.stdout.count("\n  ") == 10, r.stdout        # one reason per set
    finally:
        _oracle.set_conventions(0, 0, 0, 0)
