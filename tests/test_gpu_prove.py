"""-m gpu parity: whole-path proof bytes (HIP through the C ABI) == CPU oracle, and the oracle's verifier accepts them."""
import os

import pytest

pytestmark = pytest.mark.gpu

PROGRAMS = [
    ("+++>,<[>+.<-]", b"\x01"),      # brainfuck_air/mod.rs:807 test_proof
    ("+++><[>+<-]", b""),            # mod.rs:835 test_proof_no_input
    ("++[-]+.", b""),                # mod.rs:849 test_proof_jump_middle_of_program
    ("++++++++++[>+++++++>++++++++++>+++>+<<<<-]>++.>+.+++++++..+++.>++.<<+++++++++++++++.>.+++.------.--------.>+.>.", b""),  # mod.rs:821 hello world
]


def _first_divergence(a, b):
    for k in a:
        if a[k] != b.get(k):
            return k
    return None


@pytest.mark.parametrize("code,inp", PROGRAMS)
def test_proof_bytes_match_oracle(ctx, pkg, oracle, code, inp):
    got, tr = pkg.prove_brainfuck(code, inp, ctx=ctx, log_max_rows=20, with_transcript=True)
    want, otr, _ = oracle.prove(code, inp, log_max_rows=20)
    assert _first_divergence(otr, tr) is None, f"transcript diverges at {_first_divergence(otr, tr)}"
    assert got == want
    ok, err = oracle.verify(got, log_max_rows=20)
    assert ok, err


def test_proof_is_deterministic(ctx, pkg):
    a = pkg.prove_brainfuck(*PROGRAMS[0], ctx=ctx, log_max_rows=20)
    b = pkg.prove_brainfuck(*PROGRAMS[0], ctx=ctx, log_max_rows=20)
    assert a == b
