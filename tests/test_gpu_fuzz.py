"""-m gpu: randomized programs (tests/bf_fuzz.py, seeded): the device-resident proof is byte-identical to the oracle's, with
either table builder, and the product's own verifier accepts it."""
import pytest

from bf_fuzz import random_program

pytestmark = pytest.mark.gpu

SEEDS = [(s, 400) for s in range(100, 260)] + [(s, 6000) for s in range(300, 340)]   # (seed, step bound)


@pytest.mark.parametrize("seed,max_steps", SEEDS)
def test_random_program_proof_matches_oracle(ctx, pkg, oracle, seed, max_steps):
    code, inp, _ = random_program(seed, max_steps, min_steps=max_steps // 20)
    log_max_rows = max(max(oracle.log_sizes(code, inp)[0]), 8)
    want, otr, _ = oracle.prove(code, inp, log_max_rows=log_max_rows)
    got, tr = pkg.prove_brainfuck(code, inp, ctx=ctx, log_max_rows=log_max_rows, with_transcript=True)
    diverged = next((k for k in otr if otr[k] != tr.get(k)), None)
    assert diverged is None, f"{code!r}: transcript diverges at {diverged}"
    assert got == want, code
    assert pkg.verify_brainfuck(got, log_max_rows) == (True, "")
    if seed % 4 == 0:                                              # host table builders instead of the GPU ones
        ctx.set_table_builder(False)
        try:
            assert pkg.prove_brainfuck(code, inp, ctx=ctx, log_max_rows=log_max_rows) == want
        finally:
            ctx.set_table_builder(True)


@pytest.mark.with_poseidon
@pytest.mark.parametrize("seed", [401, 402, 403, 404, 405, 406])
def test_random_program_proof_matches_oracle_all_variants(ctx, pkg, oracle, seed):
    """A few seeds under every convention set including the Poseidon252MerkleChannel variant (whose CPU oracle is slow)."""
    code, inp, _ = random_program(seed, 300, min_steps=15)
    log_max_rows = max(max(oracle.log_sizes(code, inp)[0]), 8)
    want, otr, _ = oracle.prove(code, inp, log_max_rows=log_max_rows)
    got, tr = pkg.prove_brainfuck(code, inp, ctx=ctx, log_max_rows=log_max_rows, with_transcript=True)
    diverged = next((k for k in otr if otr[k] != tr.get(k)), None)
    assert diverged is None, f"{code!r}: transcript diverges at {diverged}"
    assert got == want, code
    assert pkg.verify_brainfuck(got, log_max_rows) == (True, "")


@pytest.mark.parametrize("seed", [501, 502, 503, 504, 505, 506, 507, 508])
def test_random_program_proof_matches_oracle_row_group_constraint_kernel(ctx, pkg, oracle, seed, monkeypatch):
    """The row-group constraint kernel (production: domains of 2^21 rows and more; covered at full size by the fib19 digest tests) forced
    onto small traces: whole proofs still equal the oracle's."""
    monkeypatch.setenv("BFHIP_CONSTRAINT_GROUP_MIN_LOG", "5")
    code, inp, _ = random_program(seed, 2000, min_steps=100)
    log_max_rows = max(max(oracle.log_sizes(code, inp)[0]), 8)
    want, _, _ = oracle.prove(code, inp, log_max_rows=log_max_rows)
    assert pkg.prove_brainfuck(code, inp, ctx=ctx, log_max_rows=log_max_rows) == want


def test_program_with_ten_distinct_component_sizes(ctx, pkg, oracle):
    """Found by tools/fuzz_campaign.py persistent (seed 40402, round 3): 10 of the 13 components have distinct sizes, i.e. 10 composition
    accumulators merged by ONE k_accumulate_sizes launch — a first version of that kernel's argument block held 9."""
    code, inp, _ = random_program(40402, 6000, min_steps=300)
    sizes = oracle.log_sizes(code, inp)[0]
    assert len(set(sizes)) >= 10
    lmr = max(sizes)
    want, _, _ = oracle.prove(code, inp, log_max_rows=lmr)
    assert pkg.prove_brainfuck(code, inp, ctx=ctx, log_max_rows=lmr) == want
