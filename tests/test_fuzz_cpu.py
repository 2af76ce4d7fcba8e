"""CPU: randomized programs (tests/bf_fuzz.py, seeded). The oracle proves and verifies each, the step count agrees with an
independent Python interpreter, and the product's host-side VM and table builders (C ABI, no GPU) agree with the oracle's."""
import numpy as np
import pytest

from bf_fuzz import random_program
from test_oracle_golden import Host

SEEDS = list(range(8))


@pytest.fixture(scope="module")
def host(pkg):
    return Host(pkg)


@pytest.mark.parametrize("seed", SEEDS)
def test_random_program_oracle_proves_and_host_side_agrees(oracle, host, seed):
    code, inp, steps = random_program(seed)
    out_o, tr_o = oracle.run(code, inp)
    out_h, tr_h = host.run(code, inp)
    assert tr_o.shape[0] == steps + 1                              # the trace ends with the halted state (machine.rs:141-238)
    assert out_o == out_h and np.array_equal(tr_o, tr_h)
    words = oracle.compile(code)
    for comp in range(13):
        assert np.array_equal(host.table(tr_h, words, comp), oracle.table(code, inp, comp)), comp
    log_max_rows = max(oracle.log_sizes(code, inp)[0])
    js, _, _ = oracle.prove(code, inp, log_max_rows=log_max_rows)
    ok, err = oracle.verify(js, log_max_rows=log_max_rows)
    assert ok, err
