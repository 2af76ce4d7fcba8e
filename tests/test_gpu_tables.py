"""-m gpu: the GPU table builders (tables.hip) produce exactly the host builders' row-granular columns
(which the CPU suite pins to the reference's golden rows), for every component, on ragged and edge-case programs."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PROGS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "programs")

CASES = [
    ("+", b""), (",.", b"A"), ("[-]", b""), ("+[-]", b""), (">>>><<<<", b""), ("++[>++[>+<-]<-]>>.", b""), (",[.,]", b"ab\x00"),
    ("+>,<[>+.<-]", b"\x01"), ("file:hello_kakarot.bf", b""), ("file:collatz.bf", b"7\n"), ("file:a-bc.bf", b"a"), ("file:loop.bf", b""), ("file:fib19.bf", b""),
]


def _code(c):
    return open(os.path.join(PROGS, c[5:])).read() if c.startswith("file:") else c


@pytest.fixture(scope="module")
def big_ctx(pkg):
    c = pkg.Context(0, max_log_domain=26)
    yield c
    c.close()


@pytest.mark.parametrize("code,inp", CASES, ids=[c[0] for c in CASES])
def test_gpu_tables_equal_host_tables(pkg, oracle, big_ctx, code, inp):
    code = _code(code)
    big_ctx.set_table_builder(True)
    tg = pkg.Trace(big_ctx, code, inp)
    big_ctx.set_table_builder(False)
    th = pkg.Trace(big_ctx, code, inp)
    big_ctx.set_table_builder(True)
    try:
        assert tg.log_sizes == th.log_sizes == oracle.log_sizes(code, inp)[0]
        ncols = [8, 8, 4, 9, 13, 13, 11, 11, 11, 11, 11, 11, 7]
        for comp in range(13):
            for col in range(ncols[comp]):
                a, b = tg.column(comp, col), th.column(comp, col)
                assert np.array_equal(a, b), f"component {comp} column {col}: first diff at row {int(np.argmax(a != b))}"
    finally:
        tg.close(); th.close()


def test_proof_identical_with_either_table_builder(pkg, oracle, big_ctx):
    code, inp = _code("file:hello_kakarot.bf"), b""
    big_ctx.set_table_builder(False)
    a = pkg.prove_brainfuck(code, inp, ctx=big_ctx, log_max_rows=17)
    big_ctx.set_table_builder(True)
    b = pkg.prove_brainfuck(code, inp, ctx=big_ctx, log_max_rows=17)
    assert a == b
    want, _, _ = oracle.prove(code, inp, log_max_rows=17)
    assert b == want


def _golden_tables():
    import json
    v = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))
    return [t for t in v["tables"] if "code" in t]


@pytest.mark.parametrize("v", _golden_tables(), ids=lambda v: f"component{v['component']}:{v['code']}")
def test_gpu_tables_match_the_reference_vectors(pkg, big_ctx, v):
    """The device table builders against the rows the reference's own unit tests pin (tests/golden/reference_vectors.json: processor,
    left, jump-if-not-zero, instruction x2, program, end-of-execution) — the same vectors that pin the oracle and the host builders."""
    big_ctx.set_table_builder(True)
    t = pkg.Trace(big_ctx, v["code"], bytes(v["input"]))
    try:
        want = np.array(v["expected"], dtype=np.uint32)
        got = np.stack([t.column(v["component"], c) for c in range(want.shape[1])], axis=1)
        assert got.shape[0] >= want.shape[0]
        assert np.array_equal(got[: want.shape[0]], want)
        if got.shape[0] > want.shape[0]:      # fixtures of the per-instruction tables list the real rows only; the rest is padding (d = 1)
            assert v["component"] in (4, 5, 6, 7, 8, 9, 10, 11)
    finally:
        t.close()
