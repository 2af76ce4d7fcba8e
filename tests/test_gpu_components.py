"""-m gpu parity of the per-component C-ABI operations against the oracle on the same inputs:
bfhip_logup_generate (LogupTraceGenerator), bfhip_eval_constraints (ComponentProver::evaluate_constraint_quotients_on_domain) and
bfhip_accumulate_quotients (QuotientOps::accumulate_quotients). Bit-exact (integer field arithmetic)."""
import numpy as np
import pytest

from conftest import splitmix_column, P

pytestmark = pytest.mark.gpu

N_COMPONENTS = 13
ALL_OPS = ("+++>,<[>+.<-]", b"\x01")       # brainfuck_air/mod.rs:807 — touches all 8 instructions
HELLO = ("++++++++++[>+++++++>++++++++++>+++>+<<<<-]>++.>+.+++++++..+++.>++.<<+++++++++++++++.>.+++.------.--------.>+.>.", b"")
NAMES = ["memory", "instruction", "program", "processor", "jnz", "jz", "input", "left", "minus", "output", "plus", "right", "end_of_execution"]


def _elems(seed):
    e = splitmix_column(seed, 24)
    e[e == 0] = 1
    return e.tolist()


def _logup_gpu(ctx, comp, rows, elems):
    """rows: (n_main, M). Returns (list of output arrays as the C ABI lays them out, claimed)."""
    n_main, M = rows.shape
    log = int(np.log2(M)) + 4
    n_logup = 3 if comp == 3 else 1
    src = [ctx.upload(rows[j]) for j in range(n_main)]
    sizes = [M] * (4 * (n_logup - 1)) + [16 * M] * 4
    dst = [ctx.malloc(4 * n) for n in sizes]
    claimed = ctx.logup_generate(comp, log, src, elems, dst)
    out = [ctx.download(p, n) for p, n in zip(dst, sizes)]
    for p in src + dst:
        ctx.free(p)
    return out, claimed


@pytest.mark.parametrize("prog", [ALL_OPS, HELLO], ids=["all_ops", "hello"])
def test_logup_generate_matches_oracle(ctx, oracle, prog):
    code, inp = prog
    elems = _elems(31)
    total = np.zeros(4, dtype=object)
    for comp in range(N_COMPONENTS):
        rows = np.ascontiguousarray(oracle.table(code, inp, comp).T)        # (n_main, M) row-granular
        want, want_claimed = oracle.logup_generate(comp, rows, elems)
        got, claimed = _logup_gpu(ctx, comp, rows, elems)
        assert claimed == want_claimed, NAMES[comp]
        n_rep = len(got) - 4
        for k in range(n_rep):                                               # replicated columns come back row-granular
            assert np.array_equal(np.repeat(got[k], 16), want[k]), (NAMES[comp], k)
        for k in range(4):
            assert np.array_equal(got[n_rep + k], want[n_rep + k]), (NAMES[comp], k)
        total = (total + np.array(claimed, dtype=object)) % P
    assert not total.any()                                                   # logUp balance: the 13 claimed sums cancel (mod.rs:189-203,772)


def _lde(oracle, cols, log):
    """cols: (k, 2^log) evaluations on CanonicCoset(log) -> (k, 2^(log+1)) on CanonicCoset(log+1) (tree_builder.commit's LDE)."""
    return oracle.evaluate(oracle.interpolate(cols, log), log, log + 1)


@pytest.mark.parametrize("replicated", [True, False, "row_group_kernel"], ids=["row_granular", "full_size", "row_group_kernel"])
@pytest.mark.parametrize("prog", [ALL_OPS, HELLO], ids=["all_ops", "hello"])
def test_eval_constraints_matches_oracle(ctx, pkg, oracle, prog, replicated, monkeypatch):
    """row_granular / full_size: the per-row kernel on replicated / full-size storage (small domains); row_group_kernel: the kernel that
    evaluates the AIR once per 16 rows (k_constraints_block, used from 2^21 rows up) forced onto these small domains."""
    if replicated == "row_group_kernel":
        monkeypatch.setenv("BFHIP_CONSTRAINT_GROUP_MIN_LOG", "5")
        replicated = True
    code, inp = prog
    elems = _elems(77)
    for comp in range(N_COMPONENTS):
        rows = np.ascontiguousarray(oracle.table(code, inp, comp).T)
        n_main, M = rows.shape
        log = int(np.log2(M)) + 4
        n = 1 << (log + 1)
        inter, claimed = oracle.logup_generate(comp, rows, elems)
        main_lde = _lde(oracle, np.repeat(rows, 16, axis=1), log)
        inter_lde = _lde(oracle, inter, log)
        one_hot = np.zeros((1, 1 << log), dtype=np.uint32); one_hot[0, 0] = 1
        is_first = _lde(oracle, one_hot, log)[0]
        n_cons = [12, 11, 5, 10, 9, 9, 7, 7, 8, 8, 8, 7, 2][comp]
        coeffs = splitmix_column(500 + comp, 4 * n_cons)
        acc0 = np.stack([splitmix_column(900 + k, n) for k in range(4)])      # non-zero start: the operation accumulates
        want = oracle.eval_constraints(comp, log, is_first, main_lde, inter_lde, elems, claimed, coeffs, acc0)

        n_logup = inter.shape[0] // 4
        inter_rep = [replicated and k < 4 * (n_logup - 1) for k in range(4 * n_logup)]
        if replicated:
            assert np.array_equal(np.repeat(main_lde[:, ::16], 16, axis=1), main_lde)   # the LDE of a broadcast column is broadcast
        p_first = ctx.upload(is_first)
        p_main = [ctx.upload(np.ascontiguousarray(c[::16] if replicated else c)) for c in main_lde]
        p_inter = [ctx.upload(np.ascontiguousarray(c[::16] if r else c)) for c, r in zip(inter_lde, inter_rep)]
        p_acc = [ctx.upload(acc0[k]) for k in range(4)]
        ctx.eval_constraints(comp, log, p_first, p_main, p_inter, elems, claimed, coeffs, p_acc,
                             main_shifts=[4 if replicated else 0] * n_main, inter_shifts=[4 if r else 0 for r in inter_rep])
        got = np.stack([ctx.download(p, n) for p in p_acc])
        for p in [p_first] + p_main + p_inter + p_acc:
            ctx.free(p)
        assert np.array_equal(got, want), NAMES[comp]


def test_component_shape(pkg):
    import ctypes
    L = pkg.lib()
    a, b, c = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
    assert L.bfhip_component_shape(3, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == 0 and (a.value, b.value, c.value) == (9, 3, 10)
    assert L.bfhip_component_shape(13, None, None, None) == -1


@pytest.mark.parametrize("log", [3, 5, 9, 14])
def test_accumulate_quotients_matches_oracle(ctx, oracle, log):
    n = 1 << log
    n_cols = 9
    shifts = [0, 4, 0, 4, 4, 0, 0, 4, 0]
    cols = []
    for k in range(n_cols):
        c = splitmix_column(40 + k, max(1, n >> shifts[k]))
        cols.append(np.repeat(c, 1 << shifts[k])[:n])
    cols = np.stack(cols)
    pts = [splitmix_column(7, 8).tolist(), splitmix_column(8, 8).tolist(), splitmix_column(9, 8).tolist()]
    # columns 0..5 sampled at point 0; column 6 at points 0 and 1 (a "last logUp column"); column 7 at point 2 only; column 8 unsampled
    which = [[0], [0], [0], [0], [0], [0], [0, 1], [2], []]
    n_samples, points, values = [], [], []
    seed = 1000
    for k in range(n_cols):
        n_samples.append(len(which[k]))
        for w in which[k]:
            points += pts[w]
            values += splitmix_column(seed, 4).tolist(); seed += 1
    coeff = splitmix_column(3, 4).tolist()
    want = oracle.accumulate_quotients(log, cols, n_samples, points, values, coeff)
    p_cols = [ctx.upload(np.ascontiguousarray(cols[k][:: 1 << shifts[k]])) for k in range(n_cols)]
    p_out = [ctx.malloc(4 * n) for _ in range(4)]
    ctx.accumulate_quotients(log, p_cols, n_samples, points, values, coeff, p_out, col_shifts=shifts)
    got = np.stack([ctx.download(p, n) for p in p_out])
    for p in p_cols + p_out:
        ctx.free(p)
    assert np.array_equal(got, want)


def test_per_component_ops_reject_bad_arguments(ctx, pkg):
    with pytest.raises(pkg.BfhipError, match="unknown component"):
        ctx.logup_generate(13, 8, [], [1] * 24, [])
    with pytest.raises(pkg.BfhipError, match="LOG_N_LANES"):
        ctx.logup_generate(0, 3, [0] * 8, [1] * 24, [0] * 4)
    with pytest.raises(pkg.BfhipError, match="twiddle tree"):
        ctx.eval_constraints(0, ctx.max_log_domain, 0, [0] * 8, [0] * 4, [1] * 24, [0] * 4, [0] * 48, [0] * 4)


def _structure_vectors():
    import json, os
    v = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))
    return v["logup_structure"]


@pytest.mark.parametrize("v", _structure_vectors(), ids=lambda v: f"component{v['component']}")
def test_logup_structure_of_the_reference_interaction_tests(ctx, oracle, v):
    """bfhip_logup_generate on the tables of the reference's 7 interaction-trace tests under LookupElements::dummy(): numerators,
    denominator columns and their order (3 logUp columns for the processor) exactly as those tests write them — expected values from the
    plain-Python LogupTraceGenerator of tests/conftest.py, not from the oracle."""
    from conftest import logup_expected_dummy_elements
    rows = np.array(v["rows"], dtype=np.uint32) if "rows" in v else oracle.table(v["code"], bytes(v["input"]), v["component"])
    assert rows.shape[0] == len(v["columns"][0]["numerators"])
    want, want_claimed = logup_expected_dummy_elements(rows.tolist(), v["columns"])
    got, claimed = _logup_gpu(ctx, v["component"], np.ascontiguousarray(rows.T), [1, 0, 0, 0] * 6)
    assert list(claimed) == [want_claimed, 0, 0, 0]
    n_rep = len(got) - 4
    for k in range(len(want) - 1):                      # non-last logUp columns come back row-granular, coordinate 0 first
        assert np.array_equal(np.repeat(got[4 * k], 16), np.array(want[k], dtype=np.uint32)), f"logUp column {k}"
        assert not any(g.any() for g in got[4 * k + 1: 4 * k + 4])
    assert np.array_equal(got[n_rep], np.array(want[-1], dtype=np.uint32))
    assert not any(g.any() for g in got[n_rep + 1:])
