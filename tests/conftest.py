import ctypes
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = (1 << 31) - 1


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "single_conv: run under the default stwo conventions only (not parametrised over CONVENTIONS)")
    config.addinivalue_line("markers", "with_poseidon: also run under the Poseidon252MerkleChannel variant")


# Byte-level stwo conventions (include/bfhip.h `bfhip_conventions`: merkle_node_hash, mix_u64, logup_mask_order). Every test that uses the
# `oracle` and/or `ctx` fixture runs once per entry, with the oracle (process-wide switch) and the context set to the same values:
#   stwo    = defaults (zero-state raw-compress Merkle nodes, raw-compress mix_u64, mask order [0, -1])
#   rfc7693 = only the Merkle node hash flipped to the RFC 7693 form (round 1's behaviour; tests/golden has the fib19 digest of both)
#   flipped = every switch on its alternative value
#   poseidon = the Poseidon252MerkleChannel variant (BASELINE config 5; 4th field merkle_channel = 1), other switches at their defaults
CONVENTIONS = {"stwo": (0, 0, 0, 0), "rfc7693": (1, 0, 0, 0), "flipped": (1, 1, 1, 0), "poseidon": (0, 0, 0, 1)}


def pytest_generate_tests(metafunc):
    if "conv" in metafunc.fixturenames:
        single = metafunc.definition.get_closest_marker("single_conv") is not None
        # the Poseidon252 variant hashes ~30x slower (GPU) and its CPU oracle far slower still: only tests marked `with_poseidon` run under it
        names = [n for n in CONVENTIONS if n != "poseidon" or metafunc.definition.get_closest_marker("with_poseidon") is not None]
        metafunc.parametrize("conv", ["stwo"] if single else names, indirect=True, scope="function")


def load_package(name="stwo_brainfuck_amd", library=None):
    """Import the hyphenated package directory `stwo-brainfuck_amd/` under the module name stwo_brainfuck_amd.
    library: bind this copy of the mirror to another build of the library (the mirror reads BFHIP_LIBRARY when it is imported)."""
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "stwo-brainfuck_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    saved = os.environ.get("BFHIP_LIBRARY")
    if library:
        os.environ["BFHIP_LIBRARY"] = library
    try:
        spec.loader.exec_module(mod)
    finally:
        if library:
            if saved is None:
                os.environ.pop("BFHIP_LIBRARY", None)
            else:
                os.environ["BFHIP_LIBRARY"] = saved
    return mod


TESTHOOKS_LIBRARY = os.path.join(ROOT, "stwo-brainfuck_amd", "libbfhip_testhooks.so")


class Oracle:
    """ctypes view of oracle/libbforacle.so — the CPU restatement used ONLY as the checker."""

    def __init__(self):
        path = os.path.join(ROOT, "oracle", "libbforacle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        self.L = ctypes.CDLL(path)
        self.L.orc_last_error.restype = ctypes.c_char_p
        self.L.orc_channel_new.restype = ctypes.c_void_p
        self.L.orc_channel_grind.restype = ctypes.c_uint64
        # OpenMP over every hardware thread of a large host is slower than a few threads for these sizes (fork/join per loop)
        self.L.orc_set_threads(min(os.cpu_count() or 1, 16))

    def _chk(self, rc):
        if rc < 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        return rc

    def set_conventions(self, merkle_node_hash=0, mix_u64=0, logup_mask_order=0, merkle_channel=0):
        self._chk(self.L.orc_set_conventions(merkle_node_hash, mix_u64, logup_mask_order, merkle_channel))

    def hash_node(self, left, right, values):
        """Blake2sMerkleHasher::hash_node under the current convention. left/right: 32-byte strings or None; values: u32 list."""
        vals = np.ascontiguousarray(values, dtype=np.uint32)
        out = (ctypes.c_ubyte * 32)()
        self._chk(self.L.orc_hash_node(left, right, vals.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(vals.size), out))
        return bytes(out)

    def blake2s(self, data: bytes) -> bytes:
        out = (ctypes.c_ubyte * 32)()
        self.L.orc_blake2s(data, ctypes.c_size_t(len(data)), out)
        return bytes(out)

    def compile(self, code: str):
        out = np.zeros(len(code) * 2 + 4, dtype=np.uint32)
        n = ctypes.c_size_t()
        self._chk(self.L.orc_compile(code.encode(), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(out.size), ctypes.byref(n)))
        return out[: n.value].tolist()

    def run(self, code: str, inp: bytes = b"", max_rows=1 << 22):
        n_out, n_rows = ctypes.c_size_t(), ctypes.c_size_t()
        self._chk(self.L.orc_run(code.encode(), inp, ctypes.c_size_t(len(inp)), None, ctypes.c_size_t(0), ctypes.byref(n_out), None, ctypes.c_size_t(0), ctypes.byref(n_rows)))
        out = (ctypes.c_ubyte * max(1, n_out.value))()
        tr = np.zeros((n_rows.value, 7), dtype=np.uint32)
        self._chk(self.L.orc_run(code.encode(), inp, ctypes.c_size_t(len(inp)), out, ctypes.c_size_t(n_out.value), ctypes.byref(n_out), tr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n_rows.value), ctypes.byref(n_rows)))
        return bytes(out[: n_out.value]), tr

    def table(self, code: str, inp: bytes, component: int):
        nr, nc = ctypes.c_size_t(), ctypes.c_size_t()
        self._chk(self.L.orc_table(code.encode(), inp, ctypes.c_size_t(len(inp)), component, None, ctypes.c_size_t(0), ctypes.byref(nr), ctypes.byref(nc)))
        out = np.zeros((nr.value, nc.value), dtype=np.uint32)
        self._chk(self.L.orc_table(code.encode(), inp, ctypes.c_size_t(len(inp)), component, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(out.size), ctypes.byref(nr), ctypes.byref(nc)))
        return out

    def log_sizes(self, code: str, inp: bytes = b""):
        out = (ctypes.c_uint32 * 13)()
        steps = ctypes.c_uint64()
        self._chk(self.L.orc_log_sizes(code.encode(), inp, ctypes.c_size_t(len(inp)), out, ctypes.byref(steps)))
        return list(out), steps.value

    def assert_constraints(self, code, inp, component, elems=None, corrupt=None):
        if elems is None:
            elems = [1, 0, 0, 0] * 6  # LookupElements::dummy(): z = alpha = 1
        e = (ctypes.c_uint32 * 24)(*elems)
        bad_row, bad_c = ctypes.c_size_t(), ctypes.c_int(-1)
        cc, cr, cv = (-1, 0, 0) if corrupt is None else corrupt
        rc = self._chk(self.L.orc_assert_constraints(code.encode(), inp, ctypes.c_size_t(len(inp)), component, e, cc, ctypes.c_size_t(cr), ctypes.c_uint32(cv), ctypes.byref(bad_row), ctypes.byref(bad_c)))
        return rc, bad_row.value, bad_c.value

    def assert_constraints_table(self, component, rows, elems):
        """rows: (n_rows, n_main) table rows. Returns (rc, first failing storage row, constraint index, value[4])."""
        cols = np.ascontiguousarray(np.asarray(rows, dtype=np.uint32).T)
        bad_row, bad_c, val = ctypes.c_size_t(), ctypes.c_int(-1), (ctypes.c_uint32 * 4)()
        rc = self._chk(self.L.orc_assert_constraints_table(component, cols.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cols.shape[1]), (ctypes.c_uint32 * 24)(*elems),
                                                            ctypes.byref(bad_row), ctypes.byref(bad_c), val))
        return rc, bad_row.value, bad_c.value, list(val)

    def prove(self, code: str, inp: bytes = b"", log_max_rows=20):
        js, n, tr, sec = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_void_p(), ctypes.c_double()
        self._chk(self.L.orc_prove(code.encode(), inp, ctypes.c_size_t(len(inp)), log_max_rows, ctypes.byref(js), ctypes.byref(n), ctypes.byref(tr), ctypes.byref(sec)))
        s = ctypes.string_at(js, n.value)
        t = ctypes.string_at(tr).decode()
        self.L.orc_free(js)
        self.L.orc_free(tr)
        return s, dict(line.split(":") for line in t.strip().split("\n")), sec.value

    def verify(self, js: bytes, log_max_rows=20):
        err = ctypes.create_string_buffer(512)
        rc = self._chk(self.L.orc_verify(js, ctypes.c_size_t(len(js)), log_max_rows, err, ctypes.c_size_t(512)))
        return rc == 0, err.value.decode()

    def interpolate(self, cols: np.ndarray, log_size: int):
        a = np.ascontiguousarray(cols, dtype=np.uint32).copy()
        self._chk(self.L.orc_circle_interpolate(a.ctypes.data_as(ctypes.c_void_p), log_size, ctypes.c_size_t(a.shape[0])))
        return a

    def evaluate(self, coeffs: np.ndarray, log_size: int, log_eval: int):
        a = np.ascontiguousarray(coeffs, dtype=np.uint32)
        out = np.zeros((a.shape[0], 1 << log_eval), dtype=np.uint32)
        self._chk(self.L.orc_circle_evaluate(a.ctypes.data_as(ctypes.c_void_p), log_size, log_eval, ctypes.c_size_t(a.shape[0]), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    # ---- per-component operations (checkers for bfhip_logup_generate / bfhip_eval_constraints / bfhip_accumulate_quotients) ----
    def logup_generate(self, component: int, rows: np.ndarray, elems24):
        """rows: (n_main, n_rows) row-granular columns. Returns ((4 * n_logup, 16 * n_rows) full-size columns, claimed[4])."""
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        n_logup = 3 if component == 3 else 1
        out = np.zeros((4 * n_logup, 16 * rows.shape[1]), dtype=np.uint32)
        claimed = (ctypes.c_uint32 * 4)()
        self._chk(self.L.orc_logup_generate(component, rows.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(rows.shape[1]), (ctypes.c_uint32 * 24)(*[int(v) for v in elems24]),
                                            out.ctypes.data_as(ctypes.c_void_p), claimed))
        return out, list(claimed)

    def eval_constraints(self, component, log_size, is_first, main, inter, elems24, claimed4, coeffs, acc):
        """All columns full size (2^(log_size+1)); acc (4, 2^(log_size+1)) is updated in place and returned."""
        is_first = np.ascontiguousarray(is_first, dtype=np.uint32); main = np.ascontiguousarray(main, dtype=np.uint32)
        inter = np.ascontiguousarray(inter, dtype=np.uint32); acc = np.ascontiguousarray(acc, dtype=np.uint32).copy()
        coeffs = [int(v) for v in coeffs]
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self._chk(self.L.orc_eval_constraints(component, log_size, p(is_first), p(main), p(inter), (ctypes.c_uint32 * 24)(*[int(v) for v in elems24]),
                                              (ctypes.c_uint32 * 4)(*[int(v) for v in claimed4]), (ctypes.c_uint32 * len(coeffs))(*coeffs), p(acc)))
        return acc

    def accumulate_quotients(self, log_size, cols, n_samples, points, values, random_coeff4):
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        out = np.zeros((4, 1 << log_size), dtype=np.uint32)
        u = lambda v: (ctypes.c_uint32 * len(v))(*[int(x) for x in v])
        self._chk(self.L.orc_accumulate_quotients(log_size, cols.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cols.shape[0]), u(n_samples), u(points), u(values),
                                                  u(random_coeff4), out.ctypes.data_as(ctypes.c_void_p)))
        return out


@pytest.fixture(scope="session")
def _oracle():
    return Oracle()


@pytest.fixture
def conv(request, _oracle, pkg):
    """The convention set of this test run: (merkle_node_hash, mix_u64, logup_mask_order). The oracle (process-wide switch), every live
    product context, contexts created during the test and the product verifier's default are switched to it."""
    name = getattr(request, "param", "stwo")
    values = CONVENTIONS[name]
    _oracle.set_conventions(*values)
    pkg.set_default_conventions(*values)
    yield values
    _oracle.set_conventions(0, 0, 0, 0)
    pkg.set_default_conventions(0, 0, 0, 0)


@pytest.fixture
def oracle(_oracle, conv):
    return _oracle


@pytest.fixture(scope="session")
def pkg():
    lib_path = os.path.join(ROOT, "stwo-brainfuck_amd", "libbfhip.so")
    if not os.path.exists(lib_path):   # normally built by __graft_entry__.build(); hipcc cross-compiles gfx950 without a GPU
        subprocess.check_call(["make", "-j", "8", "-C", os.path.join(ROOT, "stwo-brainfuck_amd", "csrc")])
    return load_package()


@pytest.fixture(scope="session")
def hooks_pkg(pkg):
    """A second copy of the Python mirror bound to libbfhip_testhooks.so — the -DBFHIP_TEST_HOOKS build of the library (csrc/Makefile): the only
    build in which BFHIP_MAILBOX_TEST_DELAY_MS / set_mailbox(test_delay_ms) and BFHIP_RCCL_LIBRARY exist. Tests of those paths use it; everything
    else runs on the default library, which has no test hooks."""
    assert os.path.exists(TESTHOOKS_LIBRARY), "libbfhip_testhooks.so is missing: make -C stwo-brainfuck_amd/csrc"
    return load_package("stwo_brainfuck_amd_testhooks", TESTHOOKS_LIBRARY)


@pytest.fixture(scope="session")
def _ctx(pkg):
    c = pkg.Context(0, max_log_domain=24)
    yield c
    c.close()


@pytest.fixture
def ctx(_ctx, conv):
    return _ctx


# ---- independent restatement of the two Merkle node-hash conventions (pure Python; checks the oracle and, through it, the kernels) ----
_B2S_IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
_B2S_SIGMA = [
    [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
    [11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4], [7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8],
    [9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13], [2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9],
    [12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11], [13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10],
    [6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5], [10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0]]


def py_blake2s_compress(h, m, t0=0, t1=0, f0=0, f1=0):
    """RFC 7693 section 3.2 compression function F for Blake2s (what stwo's blake2s_ref::compress(h, m, t0, t1, f0, f1) computes).
    h: 8 words, m: 16 words. Pinned against hashlib in tests/test_oracle_math.py::test_python_compress_reproduces_hashlib."""
    M = 0xFFFFFFFF
    rot = lambda x, r: ((x >> r) | (x << (32 - r))) & M
    v = list(h) + list(_B2S_IV)
    v[12] ^= t0; v[13] ^= t1; v[14] ^= f0; v[15] ^= f1

    def g(a, b, c, d, x, y):
        v[a] = (v[a] + v[b] + x) & M; v[d] = rot(v[d] ^ v[a], 16); v[c] = (v[c] + v[d]) & M; v[b] = rot(v[b] ^ v[c], 12)
        v[a] = (v[a] + v[b] + y) & M; v[d] = rot(v[d] ^ v[a], 8); v[c] = (v[c] + v[d]) & M; v[b] = rot(v[b] ^ v[c], 7)

    for s in _B2S_SIGMA:
        g(0, 4, 8, 12, m[s[0]], m[s[1]]); g(1, 5, 9, 13, m[s[2]], m[s[3]]); g(2, 6, 10, 14, m[s[4]], m[s[5]]); g(3, 7, 11, 15, m[s[6]], m[s[7]])
        g(0, 5, 10, 15, m[s[8]], m[s[9]]); g(1, 6, 11, 12, m[s[10]], m[s[11]]); g(2, 7, 8, 13, m[s[12]], m[s[13]]); g(3, 4, 9, 14, m[s[14]], m[s[15]])
    return [h[i] ^ v[i] ^ v[i + 8] for i in range(8)]


def py_hash_node(node_conv, left, right, values):
    """Blake2sMerkleHasher::hash_node restated for both conventions. left/right: bytes(32) or None; values: iterable of u32."""
    import hashlib
    import struct
    values = [int(v) for v in values]
    if node_conv == 1:      # RFC 7693 hash of left || right || LE values
        return hashlib.blake2s((left + right if left is not None else b"") + b"".join(struct.pack("<I", v) for v in values)).digest()
    state = [0] * 8         # stwo: zero state, raw compressions, column words zero padded to a multiple of 16
    if left is not None:
        state = py_blake2s_compress(state, list(struct.unpack("<16I", left + right)))
    rem = 15 - ((len(values) + 15) % 16)
    padded = values + [0] * rem
    for o in range(0, len(padded), 16):
        state = py_blake2s_compress(state, padded[o:o + 16])
    return struct.pack("<8I", *state)


def splitmix_column(seed: int, n: int) -> np.ndarray:
    """values = splitmix64(seed + i) mod P (BASELINE.md synthetic-input rule)."""
    x = (np.arange(n, dtype=np.uint64) + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
    x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return (x % np.uint64(P)).astype(np.uint32)


def logup_expected_dummy_elements(rows, columns):
    """Pure-Python restatement of LogupTraceGenerator (write_frac / finalize_col / finalize_last) for the reference's
    interaction-trace tests, which run under LookupElements::dummy() (z = 1, every alpha power 1): combine(values) = sum(values) - 1 is a
    base-field value, so every fraction lies in M31 and only coordinate 0 of a logUp column is non-zero.
    rows: table rows (lists of main-column values); columns: [{"denominator_columns": [...], "numerators": [...]}, ...] as the reference test
    writes them. Returns (one full-size M31 column of 16 * n_rows cells per logUp column, claimed sum). A table row occupies the 16 cells
    16 r .. 16 r + 15 of the bit-reversed circle-domain order (PackedM31 broadcast, memory/table.rs:95-104); a column before the last holds
    the running sum over the logUp columns per row; the last one is prefix-summed in COSET order."""
    M = len(rows)
    n = 16 * M
    log = n.bit_length() - 1
    run = [0] * M
    out = []
    for k, col in enumerate(columns):
        for r in range(M):
            den = (sum(rows[r][c] for c in col["denominator_columns"]) - 1) % P
            run[r] = (run[r] + (col["numerators"][r] % P) * pow(den, P - 2, P)) % P
        if k + 1 < len(columns):
            out.append([run[s >> 4] for s in range(n)])
    def bitrev(i, bits):
        return int(format(i, "0%db" % bits)[::-1], 2) if bits else 0
    last, acc = [0] * n, 0
    for i in range(n):                                   # coset position i -> circle-domain index -> bit-reversed storage cell
        idx = i // 2 if i % 2 == 0 else n - (i + 1) // 2
        s = bitrev(idx, log)
        acc = (acc + run[s >> 4]) % P
        last[s] = acc
    out.append(last)
    return out, acc
