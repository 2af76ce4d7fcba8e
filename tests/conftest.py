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


def load_package():
    """Import the hyphenated package directory `stwo-brainfuck_amd/` under the module name stwo_brainfuck_amd."""
    name = "stwo_brainfuck_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "stwo-brainfuck_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


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
def oracle():
    return Oracle()


@pytest.fixture(scope="session")
def pkg():
    lib_path = os.path.join(ROOT, "stwo-brainfuck_amd", "libbfhip.so")
    if not os.path.exists(lib_path):   # normally built by __graft_entry__.build(); hipcc cross-compiles gfx950 without a GPU
        subprocess.check_call(["make", "-j", "8", "-C", os.path.join(ROOT, "stwo-brainfuck_amd", "csrc")])
    return load_package()


@pytest.fixture(scope="session")
def ctx(pkg):
    c = pkg.Context(0, max_log_domain=24)
    yield c
    c.close()


def splitmix_column(seed: int, n: int) -> np.ndarray:
    """values = splitmix64(seed + i) mod P (BASELINE.md synthetic-input rule)."""
    x = (np.arange(n, dtype=np.uint64) + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
    x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return (x % np.uint64(P)).astype(np.uint32)
