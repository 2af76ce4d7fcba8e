"""CPU: the SIMD mode of the CPU port (oracle/simd_port.cpp, orc_set_simd(1): bench.py's cpu_baseline, the stand-in for the reference's
SimdBackend + rayon path — README.md:23-36, bin/brainfuck_prover.rs:137-139) against the scalar checker code: the AVX-512 circle FFT / iFFT,
Merkle layer and quotient rows must give the scalar results word for word, and the proofs of the reference's programs must be the same bytes."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT, P

PROGS = os.path.join(ROOT, "tests", "golden", "programs")
# the reference's bundled programs that fit the CPU suite's time (brainfuck_programs/*.bf; fib19 is the GPU suite's and bench.py's) + the
# reference's own end-to-end case (brainfuck_air/mod.rs:807)
PROGRAMS = [("a-bc.bf", b"a"), ("hello1.bf", b""), ("hello2.bf", b""), ("hello3.bf", b""), ("hello4.bf", b""), ("hello_kakarot.bf", b""), ("loop.bf", b"")]


@pytest.fixture
def simd(oracle):
    if not oracle.L.orc_simd_available():
        pytest.skip("the host has no AVX-512: the SIMD mode stays off (bench.py then reports the scalar port)")
    yield oracle
    oracle.L.orc_set_simd(0)


@pytest.mark.parametrize("log_size", [5, 6, 7, 8, 9, 10, 11, 13, 16])
def test_simd_transforms_equal_scalar(simd, log_size):
    rng = np.random.default_rng(log_size)
    cols = rng.integers(0, P, size=(3, 1 << log_size), dtype=np.uint32)
    cols[0, :4] = [0, P - 1, 1, 0]
    simd.L.orc_set_simd(0)
    coeffs, lde = simd.interpolate(cols, log_size), None
    lde = simd.evaluate(coeffs, log_size, log_size + 1)
    assert simd.L.orc_set_simd(1) == 1
    assert np.array_equal(simd.interpolate(cols, log_size), coeffs)
    assert np.array_equal(simd.evaluate(coeffs, log_size, log_size + 1), lde)


@pytest.mark.parametrize("n_cols_per_layer", [(4, 0, 4), (0, 1, 0), (16, 17, 1), (40, 3, 33)])
def test_simd_merkle_equals_scalar(simd, n_cols_per_layer):
    rng = np.random.default_rng(sum(n_cols_per_layer))
    logs = [10] * n_cols_per_layer[0] + [9] * n_cols_per_layer[1] + [6] * n_cols_per_layer[2] + [3]
    cols = [rng.integers(0, P, size=1 << l, dtype=np.uint32) for l in logs]
    ptrs = (ctypes.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
    la = np.array(logs, dtype=np.uint32)
    roots = []
    for mode in (0, 1):
        simd.L.orc_set_simd(mode)
        root = (ctypes.c_ubyte * 32)()
        layers = np.zeros(32 * (2 << max(logs)), dtype=np.uint8)
        assert simd.L.orc_merkle_commit(ptrs, la.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(cols)), root, layers.ctypes.data_as(ctypes.c_void_p)) == 0
        roots.append((bytes(root), layers.tobytes()))
    assert roots[0] == roots[1]


def _same_proof(simd, name, inp):
    code = open(os.path.join(PROGS, name)).read() if name.endswith(".bf") else name
    lmr = max(max(simd.log_sizes(code, inp)[0]), 8)
    simd.L.orc_set_simd(0)
    want, _, _ = simd.prove(code, inp, log_max_rows=lmr)
    assert simd.L.orc_set_simd(1) == 1
    got, _, _ = simd.prove(code, inp, log_max_rows=lmr)
    assert got == want
    assert simd.verify(got, lmr)[0]


@pytest.mark.single_conv
@pytest.mark.parametrize("name,inp", PROGRAMS + [("+++>,<[>+.<-]", b"\x01")])
def test_simd_proof_equals_scalar_proof(simd, name, inp):
    _same_proof(simd, name, inp)


@pytest.mark.with_poseidon
@pytest.mark.parametrize("name,inp", [("a-bc.bf", b"a"), ("+++>,<[>+.<-]", b"\x01")])
def test_simd_mode_leaves_the_other_conventions_alone(simd, conv, name, inp):
    """Under the Poseidon252 channel (and the RFC 7693 node hash) the Merkle layers stay on the scalar code; the transforms and quotients still run
    on the vector units. Same bytes."""
    if conv == (0, 0, 0, 0):
        pytest.skip("covered by test_simd_proof_equals_scalar_proof")
    _same_proof(simd, name, inp)


@pytest.mark.single_conv
def test_simd_port_under_address_and_ub_sanitizers(simd, tmp_path):
    """The SIMD mode's gathers, scatters, unaligned vector accesses and chunked parallel loops under ASan + UBSan (a separate binary: the oracle sources + a small driver),
    on a program large enough for every vector path (hello_kakarot.bf: components up to 2^17 rows). Scalar and SIMD proofs must be identical there too."""
    import subprocess
    exe = tmp_path / "oracle_simd_sanitize"
    src = [os.path.join(ROOT, "tests", "native", "oracle_simd_sanitize.cpp")] + [os.path.join(ROOT, "oracle", f) for f in ("oracle_capi.cpp", "simd_bound.cpp", "simd_port.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fopenmp", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined",
                           "-I", os.path.join(ROOT, "oracle"), "-o", str(exe)] + src + ["-lpthread"])
    r = subprocess.run([str(exe), os.path.join(PROGS, "hello_kakarot.bf"), "17"], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "identical: 1" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
