"""CPU: the product's host-side code (VM, table builders, proof JSON reader, verifier) under AddressSanitizer + UBSan.
The GPU half cannot run under a sanitizer on this pool; the host half is plain C++ headers and can. tests/native/host_sanitize.cpp is
built with g++ -fsanitize=address,undefined and fed real programs, a valid proof and ~150 mutated proofs (the verifier reads
untrusted input: it may reject, it must not read out of bounds or overflow)."""
import os
import random
import re
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "native", "host_sanitize.cpp")
PROGS = os.path.join(ROOT, "tests", "golden", "programs")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("san") / "host_sanitize")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-pthread", "-o", exe, SRC],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("g++ cannot build with sanitizers here: " + r.stderr[-300:])
    return exe


def _run(exe, *args):
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-2000:]
    return r


@pytest.mark.parametrize("name,inp", [("hello_kakarot.bf", b""), ("collatz.bf", b"7\n"), ("a-bc.bf", b"a"), ("hello3.bf", b""), ("loop.bf", b"")])
def test_vm_and_table_builders_are_clean(harness, tmp_path, name, inp):
    f = tmp_path / "in.bin"
    f.write_bytes(inp)
    r = _run(harness, "run", os.path.join(PROGS, name), str(f))
    assert r.returncode == 0 and r.stdout.startswith("steps "), r.stdout + r.stderr


@pytest.mark.single_conv      # the harness verifies under the default conventions
def test_verifier_is_clean_on_valid_and_mutated_proofs(harness, oracle, tmp_path):
    code, inp, lmr = "+++>,<[>+.<-]", b"\x01", 10
    proof, _, _ = oracle.prove(code, inp, log_max_rows=lmr)
    p = tmp_path / "proof.json"
    p.write_bytes(proof)
    assert _run(harness, "verify", str(p), str(lmr)).stdout.strip() == "ok"
    assert _run(harness, "verify", str(p), str(lmr + 1)).stdout.strip() != "ok"      # wrong LOG_MAX_ROWS
    rng = random.Random(7)
    numbers = [m.span() for m in re.finditer(rb"\d+", proof)]
    rejected = 0
    for k in range(150):
        b = bytearray(proof)
        kind = k % 5
        if kind == 0:                                  # replace a number by a huge / odd one
            s, e = rng.choice(numbers)
            b[s:e] = rng.choice([b"4294967295", b"18446744073709551615", b"99999999999999999999999", b"0", b"2147483647", b"-1"])
        elif kind == 1:                                # delete a random slice
            s = rng.randrange(len(b)); e = min(len(b), s + rng.randrange(1, 400))
            del b[s:e]
        elif kind == 2:                                # truncate
            del b[rng.randrange(1, len(b)):]
        elif kind == 3:                                # duplicate a slice (longer arrays than expected)
            s = rng.randrange(len(b)); e = min(len(b), s + rng.randrange(1, 400))
            b[s:s] = b[s:e]
        else:                                          # flip one byte
            i = rng.randrange(len(b)); b[i] = rng.randrange(32, 127)
        q = tmp_path / f"mut{k}.json"
        q.write_bytes(bytes(b))
        out = _run(harness, "verify", str(q), str(lmr)).stdout.strip()
        rejected += out != "ok"
    assert rejected >= 140          # a mutation inside a key name or whitespace may leave the proof valid; nearly all must be rejected


def test_m31_product_by_a_doubled_constant(harness):
    """m31.h m_mul_pre2 (the FFT butterflies take their twiddles doubled and fold the 64-bit product from its two words, r04) equals the plain
    product and 128-bit arithmetic on every pair of edge values and on 2 million pseudo-random pairs; so do m_add / m_sub their definitions."""
    r = _run(harness, "field", "2000000", "12345")
    assert r.returncode == 0 and r.stdout.startswith("ok 2000121"), r.stdout + r.stderr
