"""-m gpu parity: whole-path proof bytes (HIP through the C ABI) == CPU oracle, and the oracle's verifier accepts them."""
import os

import pytest

pytestmark = pytest.mark.gpu

PROGRAMS = [
    ("+++>,<[>+.<-]", b"\x01"),      # brainfuck_air/mod.rs:807 test_proof
    ("+++><[>+<-]", b""),            # mod.rs:835 test_proof_no_input
    ("++[-]+.", b""),                # mod.rs:849 test_proof_jump_middle_of_program
    ("++++++++++[>+++++++>++++++++++>+++>+<<<<-]>++.>+.+++++++..+++.>++.<<+++++++++++++++.>.+++.------.--------.>+.>.", b""),  # mod.rs:821 hello world
    ("+>,<[>+.<-]", b"\x01"),        # memory/instruction/program component.rs test_*_constraints (tests/golden air_positive)
    ("[][]+[-]", b""),               # jump_if_zero_component.rs:154 test_jump_if_zero_constraints
    ("++[-]+.", b"\x01"),            # end_of_execution/component.rs:114 (an input byte that is never read)
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
    assert pkg.verify_brainfuck(got, 20) == (True, "")          # the product's own host-side verifier (mod.rs:738)


def test_proof_is_deterministic(ctx, pkg):
    a = pkg.prove_brainfuck(*PROGRAMS[0], ctx=ctx, log_max_rows=20)
    b = pkg.prove_brainfuck(*PROGRAMS[0], ctx=ctx, log_max_rows=20)
    assert a == b


def _prog(name):
    return open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "programs", name)).read()


def test_hello_kakarot_matches_oracle(ctx, pkg, oracle):
    """BASELINE config 1 program; LOG_MAX_ROWS = its largest component (17) so the CPU oracle finishes in seconds."""
    code = _prog("hello_kakarot.bf")
    got = pkg.prove_brainfuck(code, b"", ctx=ctx, log_max_rows=17)
    want, _, _ = oracle.prove(code, b"", log_max_rows=17)
    assert got == want


def test_collatz_matches_oracle(ctx, pkg, oracle):
    code = _prog("collatz.bf")
    got = pkg.prove_brainfuck(code, b"7\n", ctx=ctx, log_max_rows=21)
    want, _, _ = oracle.prove(code, b"7\n", log_max_rows=21)
    assert got == want


def _integration_programs():
    import json
    v = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))
    return [(e["program"], bytes(e["input"]), bytes(e["expected"])) for e in v["vm_outputs"] if e["program"] != "fib19.bf"]


@pytest.mark.parametrize("name,inp,expected", _integration_programs(), ids=[p[0] for p in _integration_programs()])
def test_integration_programs_match_oracle(pkg, oracle, name, inp, expected):
    """Every program of the reference's VM integration tests (crates/brainfuck_vm/tests/integration.rs:12-104; fib19 is covered at
    full size below): the VM output is the pinned one and the device-resident proof equals the oracle's, at the smallest LOG_MAX_ROWS
    that fits the trace."""
    code = _prog(name)
    out, _ = oracle.run(code, inp)
    assert out == expected
    lmr = max(oracle.log_sizes(code, inp)[0])
    c = pkg.Context(0, max_log_domain=lmr + 2)
    try:
        got = pkg.prove_brainfuck(code, inp, ctx=c, log_max_rows=lmr)
    finally:
        c.close()
    want, _, _ = oracle.prove(code, inp, log_max_rows=lmr)
    assert got == want
    assert pkg.verify_brainfuck(got, lmr) == (True, "")


def test_fib19_full_size_proof_verifies(pkg, oracle, conv):
    """BASELINE config 2 at full size (LOG_MAX_ROWS = 24, 2^24-row memory component): the oracle's verifier accepts the HIP proof and
    rejects it after a one-word change, and the proof bytes hash to the committed digest of the oracle's own proof of this workload
    under the same conventions (digests exist for the default `stwo` set and for `rfc7693`)."""
    c = pkg.Context(0, max_log_domain=26)
    try:
        tr = pkg.Trace(c, _prog("fib19.bf"))
        assert tr.log_sizes == [24, 22, 11, 22, 19, 11, 4, 20, 19, 4, 20, 20, 4] and tr.cells == 403753616
        proof, _ = tr.prove(24)
        ok, err = oracle.verify(proof, 24)
        assert ok, err
        assert pkg.verify_brainfuck(proof, 24) == (True, "")
        bad = proof.replace(b'"proof_of_work":', b'"proof_of_work":1', 1)
        assert not oracle.verify(bad, 24)[0]
        proof2, _ = tr.prove(24)
        assert proof2 == proof        # deterministic
        # byte parity at full size: the oracle's own proof of this workload, as a committed digest (it needs minutes of CPU time;
        # tests/golden/make_fib19_proof_digest.py)
        import hashlib, json
        digests = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fib19_lmr24_oracle_proof.json")))
        want = next((d for d in digests.values() if tuple(d["conventions"]) == tuple(conv)), None)
        assert want is not None or tuple(conv) == (1, 1, 1, 0), "no committed digest for the default conventions"
        if want is not None:
            assert len(proof) == want["proof_bytes"] and hashlib.sha256(proof).hexdigest() == want["sha256"]
        tr.close()
    finally:
        c.close()


@pytest.mark.single_conv
def test_synthetic_2_to_26_row_trace_verifies(pkg, oracle):
    """BASELINE configs 3-5 family: a nested-counter program whose Memory component has 2^22 table rows = 2^26 domain rows, proved with
    the raised LOG_MAX_ROWS = 26 (2.3 * 10^9 trace cells, transforms up to 2^28 cells). No CPU proof at this size: the oracle's
    verifier and the product's own verifier must accept."""
    code = "+" * 14 + "[>" + "+" * 16000 + "[>+<-]<-]"
    c = pkg.Context(0, max_log_domain=28)
    try:
        tr = pkg.Trace(c, code, b"")
        assert max(tr.log_sizes) == 26 and tr.cells > 2 * 10**9
        proof, _ = tr.prove(26)
        tr.close()
    finally:
        c.close()
    ok, err = oracle.verify(proof, 26)
    assert ok, err
    assert pkg.verify_brainfuck(proof, 26) == (True, "")


def test_bad_program_reports_error(ctx, pkg):
    with pytest.raises(pkg.BfhipError):
        pkg.prove_brainfuck("+]", b"", ctx=ctx, log_max_rows=20)      # unbalanced bracket
    with pytest.raises(pkg.BfhipError):
        pkg.prove_brainfuck(",", b"", ctx=ctx, log_max_rows=20)       # input exhausted (machine.rs:163-169)


EDGE = [
    ("+", b""),                      # one instruction: every sub-component but `+` is an empty table (one dummy row, log_size 4)
    (",", b"\x07"),                  # input only
    (",.", b"A"),                    # echo
    ("-", b""),                      # cell wraps to P - 1 (machine.rs:189-192): mvi is a genuine field inverse
    (">>>><<<<", b""),               # pointer moves only
    ("[-]", b""),                    # jump-if-zero taken at once: the `]` table stays empty
    ("+[-]", b""),                   # both jump kinds, each once
    ("++[>++[>+<-]<-]>>.", b""),     # nested loops
    (",[.,]", b"ab\x00"),            # input-driven loop
]


@pytest.mark.with_poseidon
@pytest.mark.parametrize("code,inp", EDGE, ids=[e[0] for e in EDGE])
def test_edge_programs_match_oracle(pkg, oracle, code, inp, small_ctx):
    got = pkg.prove_brainfuck(code, inp, ctx=small_ctx, log_max_rows=12)
    want, _, _ = oracle.prove(code, inp, log_max_rows=12)
    assert got == want
    assert oracle.verify(got, 12)[0]


@pytest.fixture(scope="module")
def small_ctx(pkg):
    c = pkg.Context(0, max_log_domain=14)
    yield c
    c.close()


@pytest.mark.with_poseidon
@pytest.mark.parametrize("code,inp", [("+", b""), (",", b"\x05"), ("+[-]", b""), ("++>+<[->+<]", b"")], ids=["+", ",", "+[-]", "move"])
def test_smallest_log_max_rows(pkg, oracle, code, inp):
    """LOG_MAX_ROWS exactly the largest component (6..9 here): the smallest preprocessed tree, transforms of 2^5..2^11 cells only,
    FRI with very few layers; the context is created with the minimum twiddle tree that fits."""
    lmr = max(oracle.log_sizes(code, inp)[0])
    c = pkg.Context(0, max_log_domain=lmr + 2)
    try:
        got = pkg.prove_brainfuck(code, inp, ctx=c, log_max_rows=lmr)
    finally:
        c.close()
    assert got == oracle.prove(code, inp, log_max_rows=lmr)[0]
    assert pkg.verify_brainfuck(got, lmr) == (True, "")


def test_trace_too_large_for_log_max_rows_is_an_error(pkg, small_ctx):
    with pytest.raises(pkg.BfhipError, match="LOG_MAX_ROWS"):
        pkg.prove_brainfuck("++++++++[>++++++++<-]>[<++++>-]", b"", ctx=small_ctx, log_max_rows=6)


def test_reusing_the_preprocessed_tree_does_not_change_the_proof(pkg, oracle):
    c = pkg.Context(0, max_log_domain=22)
    try:
        a = pkg.prove_brainfuck("+++>,<[>+.<-]", b"\x01", ctx=c, log_max_rows=20)
        pkg.lib().bfhip_ctx_reuse_preprocessed(c._h, 1)
        b1 = pkg.prove_brainfuck("+++>,<[>+.<-]", b"\x01", ctx=c, log_max_rows=20)     # builds and keeps the tree
        b2 = pkg.prove_brainfuck("+++>,<[>+.<-]", b"\x01", ctx=c, log_max_rows=20)     # reuses it
        b3 = pkg.prove_brainfuck("++[-]+.", b"", ctx=c, log_max_rows=20)                # other program, same tree
        pkg.lib().bfhip_ctx_reuse_preprocessed(c._h, 0)
        assert a == b1 == b2
        assert b3 == oracle.prove("++[-]+.", b"", log_max_rows=20)[0]
    finally:
        c.close()


def test_trace_from_registers_equals_trace_from_source(pkg, oracle, ctx):
    """bfhip_trace_create_from_registers: what prove_brainfuck(&Machine) receives (mod.rs:471-473, inputs.trace() :508) — the executed
    machine's register rows and program words; no re-execution. Same proof as the source-text entry and as the oracle."""
    code, inp = "+++>,<[>+.<-]", b"\x01"
    _, rows = oracle.run(code, inp)
    t = pkg.Trace.from_registers(ctx, rows, oracle.compile(code))
    try:
        assert t.log_sizes == oracle.log_sizes(code, inp)[0]
        proof, _ = t.prove(20)
    finally:
        t.close()
    assert proof == oracle.prove(code, inp, log_max_rows=20)[0]
    with pytest.raises(pkg.BfhipError, match="EmptyTrace"):
        pkg.Trace.from_registers(ctx, rows[:0], oracle.compile(code))
    bad = rows.copy(); bad[1, 4] = (1 << 31) - 1          # not a canonical M31
    with pytest.raises(pkg.BfhipError, match="canonical"):
        pkg.Trace.from_registers(ctx, bad, oracle.compile(code))
    # register rows that are not a VM trace: clk gaps whose dummy-row total overflows 32 bits — refused by both table builders, not wrapped
    import numpy as np
    wild = np.repeat(rows[:1], 8, axis=0).copy()
    wild[:, 0] = [0, 2_000_000_000, 5, 2_100_000_000, 7, 2_147_000_000, 9, 2_147_483_000]
    wild[:, 4] = [1, 1, 2, 2, 3, 3, 4, 4]
    for on_gpu in (True, False):
        ctx.set_table_builder(on_gpu)
        try:
            with pytest.raises(pkg.BfhipError, match="2\\^28 rows"):
                pkg.Trace.from_registers(ctx, wild, oracle.compile(code))
        finally:
            ctx.set_table_builder(True)


def test_ram_size_option(pkg, ctx):
    """MachineBuilder::with_ram_size (machine.rs:56-60, `--ram-size` of bin/brainfuck_prover.rs): cell 30000 is out of range for the
    default 30000-cell RAM and fine with 30001 cells; the proof equals the one built from the same machine's register rows."""
    import ctypes
    import numpy as np
    code = ">" * 30000 + "+."
    with pytest.raises(pkg.BfhipError):
        pkg.Trace(ctx, code, b"")
    t = pkg.Trace(ctx, code, b"", ram_size=30001)
    try:
        proof, _ = t.prove(20)
    finally:
        t.close()
    L = pkg.lib()
    n_out, n_rows = ctypes.c_size_t(), ctypes.c_size_t()
    assert L.bfhip_host_run_ram(code.encode(), b"", ctypes.c_size_t(0), ctypes.c_size_t(30001), None, ctypes.c_size_t(0), ctypes.byref(n_out), None, ctypes.c_size_t(0), ctypes.byref(n_rows)) == 0
    rows = np.zeros((n_rows.value, 7), dtype=np.uint32)
    out = (ctypes.c_ubyte * 1)()
    assert L.bfhip_host_run_ram(code.encode(), b"", ctypes.c_size_t(0), ctypes.c_size_t(30001), out, ctypes.c_size_t(1), ctypes.byref(n_out), rows.ctypes.data_as(ctypes.c_void_p),
                                ctypes.c_size_t(n_rows.value), ctypes.byref(n_rows)) == 0
    assert bytes(out) == b"\x01" and rows[-1, 4] == 30000
    words = np.zeros(len(code) + 4, dtype=np.uint32); n = ctypes.c_size_t()
    assert L.bfhip_host_compile(code.encode(), words.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(words.size), ctypes.byref(n)) == 0
    t2 = pkg.Trace.from_registers(ctx, rows, words[: n.value])
    try:
        assert t2.prove(20)[0] == proof
    finally:
        t2.close()
    assert pkg.verify_brainfuck(proof, 20) == (True, "")


def test_reuse_preprocessed_survives_a_convention_change(pkg, oracle):
    """The kept preprocessed tree is keyed on what it was built under (it used to be keyed on LOG_MAX_ROWS only)."""
    c = pkg.Context(0, max_log_domain=22)
    try:
        pkg.lib().bfhip_ctx_reuse_preprocessed(c._h, 1)
        code, inp = "++[-]+.", b""
        a = pkg.prove_brainfuck(code, inp, ctx=c, log_max_rows=20)
        cur = c.get_conventions()
        other = (1 - cur[0], cur[1], cur[2])
        c.set_conventions(*other)
        b = pkg.prove_brainfuck(code, inp, ctx=c, log_max_rows=20)          # must rebuild, not reuse the tree hashed the other way
        assert pkg.verify_brainfuck(b, 20, conventions=other) == (True, "")
        c.set_conventions(*cur)
        assert pkg.prove_brainfuck(code, inp, ctx=c, log_max_rows=20) == a == oracle.prove(code, inp, log_max_rows=20)[0]
        # the MerkleChannel is part of the key too — a Blake2s tree must not serve a Poseidon252 proof, nor the reverse
        pos = (cur[0], cur[1], cur[2], 1 - cur[3])
        c.set_conventions(*pos)
        p1 = pkg.prove_brainfuck(code, inp, ctx=c, log_max_rows=20)
        assert pkg.verify_brainfuck(p1, 20, conventions=pos) == (True, "")
        assert pkg.prove_brainfuck(code, inp, ctx=c, log_max_rows=20) == p1      # second proof under the same hasher: cache hit, same bytes
        c.set_conventions(*cur)
        assert pkg.prove_brainfuck(code, inp, ctx=c, log_max_rows=20) == a
        pkg.lib().bfhip_ctx_reuse_preprocessed(c._h, 0)
    finally:
        c.close()


def test_log_max_rows_below_lanes_is_an_error(pkg, ctx):
    with pytest.raises(pkg.BfhipError, match="LOG_N_LANES"):
        pkg.prove_brainfuck("+", b"", ctx=ctx, log_max_rows=3)


def test_poseidon252_variant_on_a_2_to_22_row_trace(pkg, _oracle):
    """BASELINE config 5 family (Poseidon252MerkleChannel) at a size no CPU prover reaches in test time: a synthetic nested-counter trace
    whose Memory component has 2^22 domain rows, LOG_MAX_ROWS 22. Both verifiers must accept, tampering must be rejected, and the same
    trace under the Blake2s channel gives a different (Blake2s) proof."""
    code = "+" * 14 + "[>" + "+" * 1000 + "[>+<-]<-]"
    pkg.set_default_conventions(0, 0, 0, 1)
    _oracle.set_conventions(0, 0, 0, 1)
    try:
        c = pkg.Context(0, max_log_domain=24)
        try:
            tr = pkg.Trace(c, code, b"")
            assert max(tr.log_sizes) == 22
            proof, _ = tr.prove(22)
            assert proof == tr.prove(22)[0]
            assert b'"commitments":["0x' in proof
            assert pkg.verify_brainfuck(proof, 22) == (True, "")
            ok, err = _oracle.verify(proof, 22)
            assert ok, err
            bad = proof.replace(b'"proof_of_work":', b'"proof_of_work":1', 1)
            assert not pkg.verify_brainfuck(bad, 22)[0] and not _oracle.verify(bad, 22)[0]
            assert not pkg.verify_brainfuck(proof, 22, conventions=(0, 0, 0, 0))[0]      # a Blake2s verifier rejects the felt252 hashes
            tr.close()
        finally:
            c.close()
    finally:
        pkg.set_default_conventions(0, 0, 0, 0)
        _oracle.set_conventions(0, 0, 0, 0)


@pytest.mark.single_conv
def test_poseidon252_variant_on_a_2_to_26_row_trace(pkg, _oracle):
    """BASELINE config 5 at its own size on the one GPU of the test box: the 2^26-domain-row synthetic trace (bench.py's sweep program) under the
    Poseidon252MerkleChannel — one proof (~4.5 s, 3.4 G Hades permutations), both verifiers accept, and the SHA-256 equals the one bench.py has
    reported since round 2 (`poseidon252.proof_sha256`: the field code was rewritten on 29-bit limbs in round 3, the bytes must not move)."""
    import hashlib
    code = "+" * 14 + "[>" + "+" * 16000 + "[>+<-]<-]"
    conv = (0, 0, 0, 1)
    pkg.set_default_conventions(*conv)
    _oracle.set_conventions(*conv)
    try:
        c = pkg.Context(0, max_log_domain=28)
        try:
            tr = pkg.Trace(c, code, b"")
            assert max(tr.log_sizes) == 26
            proof, _ = tr.prove(26)
            tr.close()
        finally:
            c.close()
        assert hashlib.sha256(proof).hexdigest() == "6f4e26cc34a101f77bee86b520882855f37d9646df7f163a59307d06d9264309"
        assert pkg.verify_brainfuck(proof, 26) == (True, "")
        ok, err = _oracle.verify(proof, 26)
        assert ok, err
    finally:
        pkg.set_default_conventions(0, 0, 0, 0)
        _oracle.set_conventions(0, 0, 0, 0)


@pytest.mark.single_conv
def test_poseidon252_variant_on_a_2_to_24_row_trace_and_its_shard_group(pkg, _oracle):
    """BASELINE configs 4/5 on the one GPU of the test box: the 2^24-domain-row synthetic trace under the Poseidon252MerkleChannel (one proof,
    ~1.5 s), accepted by both verifiers, and the same trace proved by a 2-rank shard group (in-process transport) — the bytes must be equal.
    (2^26 rows run in bench.py's `poseidon252` point; the 8-GPU runs are the driver's.)"""
    import threading
    code = "+" * 14 + "[>" + "+" * 4000 + "[>+<-]<-]"
    conv = (0, 0, 0, 1)
    pkg.set_default_conventions(*conv)
    _oracle.set_conventions(*conv)
    try:
        c = pkg.Context(0, max_log_domain=26)
        try:
            tr = pkg.Trace(c, code, b"")
            assert max(tr.log_sizes) == 24
            proof, _ = tr.prove(24)
            tr.close()
        finally:
            c.close()
        assert pkg.verify_brainfuck(proof, 24) == (True, "")
        ok, err = _oracle.verify(proof, 24)
        assert ok, err
        group = pkg.LocalGroup(2)
        ctxs = [pkg.Context(0, max_log_domain=26) for _ in range(2)]
        out, errors = [None, None], []

        def run(r):
            try:
                ctxs[r].join_local_group(group, r)
                t = pkg.Trace(ctxs[r], code, b"")
                out[r] = t.prove(24)[0]
                t.close()
            except Exception as e:
                errors.append(e)
        th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
        for cx in ctxs:
            cx.leave_group(); cx.close()
        group.close()
        assert not errors, errors
        assert out[0] == proof and out[1] == proof
    finally:
        pkg.set_default_conventions(0, 0, 0, 0)
        _oracle.set_conventions(0, 0, 0, 0)


def test_bfprove_tool_prove_then_verify(tmp_path):
    """tools/bfprove.py: the prove / verify sub-commands of the reference's bin/brainfuck_prover.rs over the C ABI, proof file in between."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "proof.json"
    inp = tmp_path / "in.bin"; inp.write_bytes(b"\x01")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bfprove.py"), "prove", "--code", "+++>,<[>+.<-]", "--input-file", str(inp), "--output", str(out),
                        "--log-max-rows", "16"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bfprove.py"), "verify", str(out), "--log-max-rows", "16"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Proof verified" in r.stdout, r.stdout + r.stderr
    out.write_bytes(out.read_bytes().replace(b'"proof_of_work":', b'"proof_of_work":1', 1))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bfprove.py"), "verify", str(out), "--log-max-rows", "16"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1


def test_bench_line_contract(tmp_path):
    """bench.py prints ONE JSON line with the contract's keys; the timed proof is checked against the committed digest (parity_checked),
    the roofline is the VALU one with a measured compression count, and the sweep points verify."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-baseline", "sample", "--sweep-logs", "20,22", "--sweep-steps", "1",
                        "--poseidon-log", "20"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert d["parity_checked"] is True and d["parity"]["own_verifier_accepts"] is True
    assert abs(d["value"] - d["config"]["cells_per_proof"] * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["kernel"] == "k_merkle_layer" and rf["bound"] == "valu" and 0.5 < rf["frac"] < 1.0 and 0.95 * 674228124 < rf["compressions_per_proof"] <= 674228124 and rf["compressions_per_proof_all_merkle_kernels"] == 674246651
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["hbm"]["peak"] == 8000.0
    assert d["fft"]["algorithmic_GBps"] > 0 and any(k.startswith("k_fft_strided7") for k in d["fft"]["kernels"])
    assert [p["log_domain_rows"] for p in d["sweep"]] == [20, 22] and all(p["verified"] for p in d["sweep"])
    assert d["config"]["headline_2^22"]["cells_per_s"] > 0 and all(len(p["proof_sha256"]) == 64 for p in d["sweep"])
    assert d["poseidon252"]["verified"] is True and d["poseidon252"]["conventions"] == [0, 0, 0, 1] and d["poseidon252"]["log_domain_rows"] == 20
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port-simd", "port") and cb["value"] > 0 and "sample" in cb
    assert cb["cores"] == cb["cores_effective"] >= 1 and cb["threads"] <= max(cb["cores_effective"], 1) and "gpu_over_port" not in cb      # effective cores, no ratio against the scalar port
    # r05: both roofline fractions ride in the line (frac_rocprof is null with its reason when the committed summary is stale), proofs in flight at three sizes
    assert "frac_rocprof" in rf and "frac_rocprof_source" in rf and len(rf["kernel_sources_sha256"]) == 64
    pl = d["pipelined"]
    for w in ("fib19", "2^22_rows", "2^20_rows"):
        assert pl[w]["in_flight_2"]["same_proof_as_1"] and pl[w]["in_flight_3"]["all_same_proof"] and pl[w]["in_flight_1"]["ms_per_proof"] > 0, pl[w]
    assert d["metric_point"]["pipelined"]["in_flight_2"]["ms_per_proof"] > 0
    assert "2^22 rows" in d["config"]["workload"] and "fib19" in d["config"]["workload"]


def test_bench_line_is_self_explaining(tmp_path):
    """the line must say whether the box was slow (roofline.sustained_clock_ghz from the library's in-run clock probe, the VALU
    fraction at THAT clock beside the one at the nominal 2.4 GHz), carry the metric's own numbers inside `config` (the object the driver's record keeps
    whole): the 2^22-row point, the same size through the library's pool from one caller thread with and without the shared preprocessed tree, the
    sweep's ms per proof — and report the FFT kernels against their VALU bound as well as against HBM."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--sweep-logs", "20,22", "--sweep-steps", "2", "--no-poseidon"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    rf = d["roofline"]
    assert 1.5 < rf["sustained_clock_ghz"] <= 2.45, rf.get("clock_probe")
    assert abs(rf["frac_at_sustained_clock"] - rf["frac"] * 2.4 / rf["sustained_clock_ghz"]) < 2e-3 and 0.6 < rf["frac_at_sustained_clock"] < 1.02
    cp = rf["clock_probe"]
    assert 0.9 < cp["probe_frac_at_its_clock"] < 1.02 and cp["register_only"]["ghz_min"] <= cp["register_only"]["ghz"] <= cp["register_only"]["ghz_max"], cp      # the register-only loop issues at its clock's peak
    mk = cp["merkle_kernel"]                                                   # the real kernel on a fixed shape beside a one-wave clock sampler
    assert mk["sampler_spanned_the_window"] is True and rf["sustained_clock_ghz"] == mk["ghz"] and 0.8 < mk["frac_of_nominal_valu_peak"] < 1.0 and 0.85 < mk["vs_builder_boxes"] < 1.1, mk
    assert isinstance(cp["device_is_slow"], bool) and (cp["device_is_slow_because"] is None) == (not cp["device_is_slow"])
    cfg = d["config"]
    mp = cfg["metric_point"]
    assert set(mp) == {"rows", "ms_per_proof", "cells_per_s", "sha256", "verified"} and mp["rows"] == "2^22" and mp["verified"] is True and len(mp["sha256"]) == 64
    assert abs(mp["cells_per_s"] - d["metric_point"]["cells"] / (mp["ms_per_proof"] * 1e-3)) / mp["cells_per_s"] < 1e-3
    assert set(cfg["sweep_ms"]) == {"2^20", "2^22"} and cfg["sweep_ms"]["2^22"] == mp["ms_per_proof"]
    b = cfg["batch"]
    for k in ("in_flight_2", "in_flight_3"):
        sh, rc = b[k]["shared_preprocessed"], b[k]["recommitted_preprocessed"]
        assert sh["same_proof_as_1"] is True and sh["ms_per_proof"] > 0 and rc["ms_per_proof"] > 0 and sh["gain_vs_1"] > 1.0, b
        assert sh["ms_per_proof"] <= rc["ms_per_proof"] * 1.03, b            # sharing the preprocessed commitment never costs
    assert b["ms_per_proof"] == min(b["in_flight_2"]["shared_preprocessed"]["ms_per_proof"], b["in_flight_3"]["shared_preprocessed"]["ms_per_proof"])
    assert "one caller thread" in b["what"].lower() or "ONE caller thread" in b["what"]
    for name, k in d["fft"]["kernels"].items():
        assert 0.0 < k["valu_frac"] < 1.0 and k["butterflies"] > 0 and 0.0 < k["moved_frac_of_hbm_peak"] < 1.0, (name, k)
    assert 0.0 < d["fft"]["valu_frac"] < 1.0


def test_clock_probe_reads_a_plausible_sustained_clock(pkg):
    """bfhip_clock_probe: d(s_memtime) / d(s_memrealtime) x 100 MHz around a register-only Blake2s loop. The clock lies between the part's floor under load
    and its 2.4 GHz maximum, and at that clock the loop issues at the integer-VALU peak (977 lane-ops per compression, 256 CUs x 64 lanes per clock)."""
    c = pkg.Context(0, max_log_domain=14)
    try:
        p = c.clock_probe(0.5)
        assert 1.2 < p["ghz_min"] <= p["ghz"] <= p["ghz_max"] <= 2.45, p
        assert 0.9 < p["G_compressions_per_s"] * 1e9 * 977 / (256 * 64 * p["ghz"] * 1e9) < 1.02, p
        with pytest.raises(pkg.BfhipError):
            c.clock_probe(0.0)
        # the same under the real Merkle kernel: a one-wave sampler on the other stream spans the window of back-to-back k_merkle_layer launches
        m = c.clock_probe_mix(0.3)
        assert m["sampler_spanned_the_window"] is True and 0.25 < m["sampler_seconds"] < 1.0 and 1.2 < m["ghz"] <= 2.45 and m["launches"] >= 64, m
        assert 0.8 < m["G_compressions_per_s"] * 1e9 * 977 / (256 * 64 * 2.4e9) < 1.0, m
        assert pkg.prove_brainfuck("+++>,<[>+.<-]", b"\x01", ctx=c, log_max_rows=12) is not None        # the context proves as before (the probe's pinned word and side stream are its own)
    finally:
        c.close()


def test_prove_entries_reject_null_arguments(pkg, ctx):
    """Null context / trace / program text: -1 and a message, not a crash."""
    import ctypes
    L = pkg.lib()
    js, n = ctypes.c_void_p(), ctypes.c_size_t()
    assert L.bfhip_prove_trace(ctx._h, None, 20, ctypes.byref(js), ctypes.byref(n), None, None) == -1 and b"null trace" in L.bfhip_last_error()
    assert L.bfhip_prove_trace(None, None, 20, ctypes.byref(js), ctypes.byref(n), None, None) == -1 and b"null context" in L.bfhip_last_error()
    assert L.bfhip_prove_brainfuck(ctx._h, None, None, ctypes.c_size_t(0), 20, ctypes.byref(js), ctypes.byref(n), None, None) == -1
    assert L.bfhip_prove_brainfuck(None, b"+", None, ctypes.c_size_t(0), 20, ctypes.byref(js), ctypes.byref(n), None, None) == -1
    t = ctypes.c_void_p()
    assert L.bfhip_trace_create(ctx._h, None, None, ctypes.c_size_t(0), ctypes.byref(t), None, None, None, None) == -1
    assert L.bfhip_trace_create(ctx._h, b"+", None, ctypes.c_size_t(0), None, None, None, None, None) == -1
    assert L.bfhip_trace_column(ctx._h, None, 0, 0, None, ctypes.c_size_t(0), ctypes.byref(n)) == -1


@pytest.mark.single_conv
def test_overlap_modes_do_not_change_the_proof(pkg, oracle):
    """bfhip_ctx_set_overlap: the two intra-proof overlaps (tree commitment on a partner stream beside the transforms of the smaller columns;
    FRI first-layer tree level by level behind the quotient launches) reorder work across streams, never bytes. collatz at LOG_MAX_ROWS 21 has
    trees of 2^22 leaves, large enough for both paths to engage."""
    code = _prog("collatz.bf")
    want, _, _ = oracle.prove(code, b"7\n", log_max_rows=21)
    c = pkg.Context(0, max_log_domain=23)
    try:
        for mask in (0, 1, 2, 3, 7, 0):          # bit 2 (shard-group exchanges) has nothing to do outside a group
            c.set_overlap(mask)
            for _ in range(2):
                assert pkg.prove_brainfuck(code, b"7\n", ctx=c, log_max_rows=21) == want, f"overlap mask {mask}"
        with pytest.raises(pkg.BfhipError):
            c.set_overlap(8)
    finally:
        c.close()


@pytest.mark.single_conv
def test_mailbox_order_does_not_change_the_proof_and_a_late_host_fails_loudly(pkg, hooks_pkg, oracle, monkeypatch):
    """csrc/mailbox.hip: for proofs with LOG_MAX_ROWS <= 21 (or with BFHIP_MAILBOX=1) the launches behind a Fiat-Shamir point are on the stream before
    the host knows the challenge (a one-workgroup kernel waits for the host's flag and copies the challenge-dependent tables); BFHIP_MAILBOX=0
    (and larger proofs) keep wait -> compute -> copy -> launch.
    Same bytes either way. A host that is later than the kernel's patience (1 ms of patience, 30 ms of test delay before every post) must end
    in an ERROR — the kernels ran on stale challenge words — never in a hang or in a proof, and the context proves correctly afterwards.
    The delay is a test hook: it exists only in libbfhip_testhooks.so (hooks_pkg); the default library refuses it."""
    code = _prog("collatz.bf")
    want, _, _ = oracle.prove(code, b"7\n", log_max_rows=21)
    for env in ("1", "0"):
        monkeypatch.setenv("BFHIP_MAILBOX", env)
        c = pkg.Context(0, max_log_domain=23)
        try:
            for _ in range(3):
                assert pkg.prove_brainfuck(code, b"7\n", ctx=c, log_max_rows=21) == want, f"BFHIP_MAILBOX={env}"
                assert c.last_proof_flags()["mailbox_order"] == (env == "1")       # the forced mode really applied (one proof in flight)
        finally:
            c.close()
    monkeypatch.setenv("BFHIP_MAILBOX", "1")
    monkeypatch.setenv("BFHIP_MAILBOX_TIMEOUT_MS", "1")
    monkeypatch.setenv("BFHIP_MAILBOX_TEST_DELAY_MS", "30")
    c = pkg.Context(0, max_log_domain=23)       # the default library: the environment's test delay does not exist there
    try:
        with pytest.raises(pkg.BfhipError, match="test-hooks build"):
            c.set_mailbox(1, 10000, 30)
        c.set_mailbox(1, 10000, -1)
        assert pkg.prove_brainfuck(code, b"7\n", ctx=c, log_max_rows=21) == want
    finally:
        c.close()
    pkg = hooks_pkg                              # from here on: the test-hooks build
    c = pkg.Context(0, max_log_domain=23)
    try:
        # the error names its cause (a late host), not the constraint mismatch the stale challenge words produce
        with pytest.raises(pkg.BfhipError, match="mailbox kernel gave up waiting for the host"):
            pkg.prove_brainfuck(code, b"7\n", ctx=c, log_max_rows=21)
        # the SAME context recovers: the timeout and the test delay are reset on the live context (Ctx::init read the environment), the stale
        # flag / stamp slots, the error words, the staging ring and the pending-mailbox count of the failed proof must not leak into the next
        c.set_mailbox(1, 10000, 0)
        for _ in range(3):
            assert pkg.prove_brainfuck(code, b"7\n", ctx=c, log_max_rows=21) == want
        c.set_mailbox(0, 0, -1)
        assert pkg.prove_brainfuck(code, b"7\n", ctx=c, log_max_rows=21) == want
        # the blocking sync policy switches the automatic mailbox mode off (and a forced one still proves the same bytes, sleeping in its waits)
        c.set_mailbox(-1, 0, -1)
        c.set_sync_policy(True)
        assert pkg.prove_brainfuck(code, b"7\n", ctx=c, log_max_rows=21) == want
        c.set_mailbox(1, 0, -1)
        assert pkg.prove_brainfuck(code, b"7\n", ctx=c, log_max_rows=21) == want
    finally:
        c.close()
    monkeypatch.delenv("BFHIP_MAILBOX_TIMEOUT_MS"); monkeypatch.delenv("BFHIP_MAILBOX_TEST_DELAY_MS")
    c = pkg.Context(0, max_log_domain=23)
    try:
        assert pkg.prove_brainfuck(code, b"7\n", ctx=c, log_max_rows=21) == want
    finally:
        c.close()


@pytest.mark.single_conv
def test_several_proofs_in_flight_in_the_mailbox_order_do_not_block_each_other(pkg, oracle, monkeypatch):
    """Four contexts of one process proving at the same time with the mailbox order forced on: their streams share the GPU's few hardware queues, and two
    spinning mailbox kernels could each sit in front of the kernels the OTHER proof's host thread waits for (seen in round 5: three 2^20-row proofs in flight ->
    'a mailbox kernel gave up waiting for the host'). Only one proof of the process holds the mailbox order at a time; the others keep the plain order. Same bytes,
    no error, well inside the kernels' patience (3 s here)."""
    import threading
    monkeypatch.setenv("BFHIP_MAILBOX", "1")
    monkeypatch.setenv("BFHIP_MAILBOX_TIMEOUT_MS", "3000")
    code = _prog("collatz.bf")
    want, _, _ = oracle.prove(code, b"7\n", log_max_rows=21)
    ctxs = [pkg.Context(0, max_log_domain=23) for _ in range(4)]
    errors, wrong = [], []

    def run(c):
        try:
            for _ in range(8):
                if pkg.prove_brainfuck(code, b"7\n", ctx=c, log_max_rows=21) != want:
                    wrong.append(1)
        except Exception as e:
            errors.append(repr(e))

    try:
        th = [threading.Thread(target=run, args=(c,)) for c in ctxs]
        [t.start() for t in th]; [t.join() for t in th]
    finally:
        for c in ctxs:
            c.close()
    assert not errors and not wrong, (errors, len(wrong))
