"""CPU: the oracle (and the product's host-side VM / table builders) against the reference's own golden vectors
(tests/golden/reference_vectors.json, hand-transcribed from the reference's unit tests) and public KATs."""
import ctypes
import json
import os

import numpy as np
import pytest

from conftest import ROOT, P

V = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))
PROGS = os.path.join(ROOT, "tests", "golden", "programs")


def prog(name):
    return open(os.path.join(PROGS, name)).read()


# ---- product host side through the C ABI (no GPU needed) -------------------------------------------------------------------------
class Host:
    def __init__(self, pkg):
        self.L = pkg.lib()

    def compile(self, code):
        out = np.zeros(2 * len(code) + 4, dtype=np.uint32)
        n = ctypes.c_size_t()
        assert self.L.bfhip_host_compile(code.encode(), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(out.size), ctypes.byref(n)) == 0
        return out[: n.value].tolist()

    def run(self, code, inp):
        n_out, n_rows = ctypes.c_size_t(), ctypes.c_size_t()
        assert self.L.bfhip_host_run(code.encode(), inp, ctypes.c_size_t(len(inp)), None, ctypes.c_size_t(0), ctypes.byref(n_out), None, ctypes.c_size_t(0), ctypes.byref(n_rows)) == 0
        out = (ctypes.c_ubyte * max(1, n_out.value))()
        tr = np.zeros((n_rows.value, 7), dtype=np.uint32)
        assert self.L.bfhip_host_run(code.encode(), inp, ctypes.c_size_t(len(inp)), out, ctypes.c_size_t(n_out.value), ctypes.byref(n_out),
                                     tr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n_rows.value), ctypes.byref(n_rows)) == 0
        return bytes(out[: n_out.value]), tr

    def table(self, trace, code_words, component):
        trace = np.ascontiguousarray(trace, dtype=np.uint32).reshape(-1, 7)
        cw = np.ascontiguousarray(code_words, dtype=np.uint32)
        nr, nc = ctypes.c_size_t(), ctypes.c_size_t()
        args = (trace.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(trace.shape[0]), cw.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cw.size), component)
        assert self.L.bfhip_host_table(*args, None, ctypes.c_size_t(0), ctypes.byref(nr), ctypes.byref(nc)) == 0
        out = np.zeros((nr.value, nc.value), dtype=np.uint32)
        assert self.L.bfhip_host_table(*args, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(out.size), ctypes.byref(nr), ctypes.byref(nc)) == 0
        return out


@pytest.fixture(scope="module")
def host(pkg):
    return Host(pkg)


def oracle_table_from_registers(oracle, trace, code_words, component):
    trace = np.ascontiguousarray(trace, dtype=np.uint32).reshape(-1, 7)
    cw = np.ascontiguousarray(code_words, dtype=np.uint32)
    nr, nc = ctypes.c_size_t(), ctypes.c_size_t()
    args = (trace.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(trace.shape[0]), cw.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cw.size), component)
    assert oracle.L.orc_table_from_registers(*args, None, ctypes.c_size_t(0), ctypes.byref(nr), ctypes.byref(nc)) == 0
    out = np.zeros((nr.value, nc.value), dtype=np.uint32)
    assert oracle.L.orc_table_from_registers(*args, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(out.size), ctypes.byref(nr), ctypes.byref(nc)) == 0
    return out


# ---- compiler / VM ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("v", V["compile"])
def test_compile_golden(oracle, host, v):
    assert oracle.compile(v["code"]) == v["expected"]
    assert host.compile(v["code"]) == v["expected"]


def test_compile_strips_whitespace(oracle, host):
    # compiler.rs:50-59 (test_whitespace)
    assert oracle.compile(" +  +> , < [> + .< - ]  ") == oracle.compile("++>,<[>+.<-]")
    assert host.compile(" +  +> , < [> + .< - ]  ") == host.compile("++>,<[>+.<-]")


@pytest.mark.parametrize("v", V["trace"])
def test_trace_golden(oracle, host, v):
    _, tr = oracle.run(v["code"], bytes(v["input"]))
    assert tr.tolist() == v["expected"]
    _, tr2 = host.run(v["code"], bytes(v["input"]))
    assert tr2.tolist() == v["expected"]


@pytest.mark.parametrize("v", V["vm_unit"], ids=lambda v: v["cite"].split("(")[1].rstrip(")"))
def test_vm_unit_cases(oracle, host, v):
    """The reference's seven single-instruction VM tests (machine.rs:291-392): the words it hands the machine are what both compilers emit,
    and the asserted fields of the final state hold for both VMs."""
    for side in (oracle, host):
        assert side.compile(v["code"]) == v["code_words"]
        out, tr = side.run(v["code"], bytes(v["input"]))
        last = dict(zip(("clk", "ip", "ci", "ni", "mp", "mv", "mvi"), tr[-1].tolist()))
        for reg, want in v.get("final", {}).items():
            assert last[reg] == want, (reg, last)
        if "ram0" in v:     # memory cell 0 = mv of the last row recorded with the pointer on it
            assert [r for r in tr.tolist() if r[4] == 0][-1][5] == v["ram0"]
        if "output" in v:
            assert list(out) == v["output"]


@pytest.mark.parametrize("v", V["vm_outputs"], ids=lambda v: v["program"])
def test_vm_outputs_golden(oracle, host, v):
    out, tr = oracle.run(prog(v["program"]), bytes(v["input"]))
    assert list(out) == v["expected"]
    out2, tr2 = host.run(prog(v["program"]), bytes(v["input"]))
    assert list(out2) == v["expected"]
    assert np.array_equal(tr, tr2)


def test_mvi_is_inverse(oracle):
    _, tr = oracle.run(prog("hello_kakarot.bf"))
    mv, mvi = tr[:, 5].astype(np.uint64), tr[:, 6].astype(np.uint64)
    assert np.all(((mv * mvi) % P)[mv != 0] == 1) and np.all(mvi[mv == 0] == 0)


# ---- tables -----------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("v", V["tables"], ids=lambda v: f"component{v['component']}")
def test_tables_golden(oracle, host, v):
    if "trace" in v:
        trace, code_words = v["trace"], v["code_words"]
    else:
        _, trace = oracle.run(v["code"], bytes(v["input"]))
        code_words = oracle.compile(v["code"])
    want = np.array(v["expected"], dtype=np.uint32)
    assert np.array_equal(oracle_table_from_registers(oracle, trace, code_words, v["component"]), want)
    assert np.array_equal(host.table(trace, code_words, v["component"]), want)


@pytest.mark.parametrize("v", V["table_errors"], ids=lambda v: f"component{v['component']}:{v['error']}")
def test_table_error_paths(oracle, host, v):
    """TraceError::EmptyTrace / InvalidEndOfExecution (the reference's empty-table and end-of-execution tests): both builders refuse."""
    trace = np.ascontiguousarray(v["trace"], dtype=np.uint32).reshape(-1, 7)
    cw = np.ascontiguousarray(v["code_words"], dtype=np.uint32)
    nr, nc = ctypes.c_size_t(), ctypes.c_size_t()
    args = (trace.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(trace.shape[0]), cw.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cw.size), v["component"])
    assert oracle.L.orc_table_from_registers(*args, None, ctypes.c_size_t(0), ctypes.byref(nr), ctypes.byref(nc)) != 0
    assert v["error"] in oracle.L.orc_last_error().decode()
    assert host.L.bfhip_host_table(*args, None, ctypes.c_size_t(0), ctypes.byref(nr), ctypes.byref(nc)) != 0
    assert v["error"] in host.L.bfhip_last_error().decode()


def _logup_rows(oracle, v):
    if "rows" in v:
        return np.array(v["rows"], dtype=np.uint32)
    return oracle.table(v["code"], bytes(v["input"]), v["component"])


DUMMY_ELEMENTS = [1, 0, 0, 0] * 6        # LookupElements::dummy(): z = 1, alpha = 1 for the three relations


@pytest.mark.parametrize("v", V["logup_structure"], ids=lambda v: f"component{v['component']}")
def test_logup_structure_of_the_reference_interaction_tests(oracle, v):
    """The 7 interaction-trace tests of the reference list, row by row, the numerator and the denominator columns each component writes
    (under LookupElements::dummy()). The oracle's logUp generation on the same table must produce exactly the columns and the claimed sum
    that a plain Python LogupTraceGenerator makes of that list."""
    from conftest import logup_expected_dummy_elements
    rows = _logup_rows(oracle, v)
    assert rows.shape[0] == len(v["columns"][0]["numerators"]), "row count of the table differs from the reference test's"
    want, want_claimed = logup_expected_dummy_elements(rows.tolist(), v["columns"])
    got, claimed = oracle.logup_generate(v["component"], np.ascontiguousarray(rows.T), DUMMY_ELEMENTS)
    assert list(claimed) == [want_claimed, 0, 0, 0]
    for k, col in enumerate(want):
        assert np.array_equal(got[4 * k], np.array(col, dtype=np.uint32)), f"logUp column {k}"
        assert not got[4 * k + 1: 4 * k + 4].any()


@pytest.mark.parametrize("v", V["air_negative"], ids=lambda v: v["cite"].split("(")[1].split(":")[0].split(")")[0])
def test_air_negative_cases_of_the_reference(oracle, host, v):
    """The Memory component's 10 negative AIR tests: same registers, same patched cells; the first constraint that fails is the one the test
    is named after, at the table row and with the value the reference's panic message quotes. Both table builders produce the table."""
    elems = [5, 1, 2, 3, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83]      # drawn elements (dummy ones give a zero denominator)
    rows = oracle_table_from_registers(oracle, v["trace"], [43], v["component"])
    assert np.array_equal(rows, host.table(v["trace"], [43], v["component"]))
    for r, c, val in v["patch"]:
        rows[r, c] = val
    rc, bad_row, bad_c, value = oracle.assert_constraints_table(v["component"], rows, elems)
    assert rc == 1
    assert bad_row // 16 == v["table_row"] and bad_c == v["constraint"] and value == [v["value"], 0, 0, 0], (bad_row, bad_c, value)


def test_memory_dummy_entries_do_not_change_the_claimed_sum(oracle):
    """memory/table.rs:886-929 (test_interaction_trace_evaluation_dummy_entries_effect): the clk-gap and padding dummies (d = 1) contribute
    numerator 0 — the claimed sum of the table with them equals the claimed sum of the real entries alone."""
    with_dummies = [[0, 43, 91, 0, 1, 43, 91, 1], [1, 43, 91, 1, 2, 91, 9, 0], [2, 91, 9, 0, 2, 91, 9, 1], [2, 91, 9, 1, 3, 91, 9, 1]]
    real_only = [[0, 43, 91, 0, 2, 91, 9, 0], [2, 91, 9, 0, 3, 91, 9, 1]]
    a = oracle.logup_generate(0, np.ascontiguousarray(np.array(with_dummies, dtype=np.uint32).T), DUMMY_ELEMENTS)[1]
    b = oracle.logup_generate(0, np.ascontiguousarray(np.array(real_only, dtype=np.uint32).T), DUMMY_ELEMENTS)[1]
    assert list(a) == list(b)


@pytest.mark.parametrize("name,inp", [("hello_kakarot.bf", b""), ("collatz.bf", b"7\n"), ("a-bc.bf", b"a"), ("loop.bf", b"")])
def test_host_tables_equal_oracle_tables(oracle, host, name, inp):
    code = prog(name)
    _, trace = oracle.run(code, inp)
    cw = oracle.compile(code)
    for comp in range(13):
        assert np.array_equal(host.table(trace, cw, comp), oracle.table(code, inp, comp)), f"component {comp}"


@pytest.mark.parametrize("v", V["log_sizes"], ids=lambda v: v["program"])
def test_log_sizes(oracle, v):
    ls, steps = oracle.log_sizes(prog(v["program"]), bytes(v["input"]))
    assert ls == v["expected"] and steps == v["steps"]


def test_empty_subcomponent_gets_one_dummy_row(oracle):
    # instructions/table.rs:293-307: 0usize.next_power_of_two() == 1 -> one dummy entry -> log_size 4
    t = oracle.table("+", b"", 6)  # no ',' in the program
    assert t.shape == (1, 11) and t[0, 7] == 1 and not t[0, :7].any()


def test_end_of_execution_requires_exactly_one_row(oracle):
    # end_of_execution/table.rs:75-77 — exactly one row with ci == 0, always the last trace row
    t = oracle.table("++", b"", 12)
    assert t.shape == (1, 7) and t[0, 2] == 0 and t[0, 0] == 2


# ---- hashing ----------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("v", V["blake2s"])
def test_blake2s_kat(oracle, v):
    assert oracle.blake2s(bytes.fromhex(v["msg_hex"])).hex() == v["digest_hex"]


def test_blake2s_matches_hashlib_on_random_lengths(oracle):
    import hashlib
    rng = np.random.default_rng(7)
    for n in [1, 31, 32, 63, 64, 65, 127, 128, 129, 1000]:
        m = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert oracle.blake2s(m) == hashlib.blake2s(m).digest()


def test_host_run_ram_size(pkg):
    """Machine::new_with_config (machine.rs:116-131): RAM size is a parameter; the default is 30000 cells (machine.rs:114)."""
    L = pkg.lib()
    code = (">" * 30000 + "+").encode()
    n_out, n_rows = ctypes.c_size_t(), ctypes.c_size_t()
    args = (None, ctypes.c_size_t(0), ctypes.byref(n_out), None, ctypes.c_size_t(0), ctypes.byref(n_rows))
    assert L.bfhip_host_run(code, b"", ctypes.c_size_t(0), *args) != 0                                   # cell 30000 is out of range
    assert L.bfhip_host_run_ram(code, b"", ctypes.c_size_t(0), ctypes.c_size_t(0), *args) != 0          # 0 = the default size
    assert L.bfhip_host_run_ram(code, b"", ctypes.c_size_t(0), ctypes.c_size_t(30001), *args) == 0
    assert n_rows.value == 30002


@pytest.mark.single_conv
def test_host_side_agrees_with_the_oracle_on_unconstrained_programs(pkg, oracle):
    """A fixed slice of tools/fuzz_vm.py: random strings over the instruction set — most are invalid or fail at run time. Compile status and
    words, run status, output, register trace and all 13 tables of the product's host side equal the oracle's."""
    import random, subprocess, sys, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_vm.py"), "8", "424242"], capture_output=True, text=True, timeout=300)
    d = json.loads(r.stdout)
    assert r.returncode == 0 and d["ok"], d["problems"]
    assert d["programs"] > 2000 and d["compile_errors"] > 100 and d["run_errors"] > 100 and d["ran"] > 300 and d["tables_compared"] > 3000


@pytest.mark.parametrize("v", V["air_positive"], ids=lambda v: v["cite"].split("(")[1].split(")")[0])
def test_air_positive_cases_of_the_reference(oracle, host, v):
    """The 13 `test_*_constraints` of the reference: the table of one component from the cited program run has 2^LOG_SIZE / 16 rows (LOG_SIZE is a
    literal of each test), both table builders agree on it, and every constraint — the logUp ones included, under LookupElements::dummy() as the
    reference draws them — vanishes on the whole trace domain."""
    inp = bytes(v["input"])
    rows = oracle.table(v["code"], inp, v["component"])
    assert 16 * rows.shape[0] == 1 << v["log_size"]
    _, trace = oracle.run(v["code"], inp)
    assert np.array_equal(rows, host.table(trace, oracle.compile(v["code"]), v["component"]))
    rc, bad_row, bad_c, value = oracle.assert_constraints_table(v["component"], rows, DUMMY_ELEMENTS)
    assert rc == 0, (bad_row, bad_c, value)
