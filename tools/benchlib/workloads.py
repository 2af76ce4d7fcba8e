"""The bench workloads, the package loader and the committed digests (see tools/benchlib/__init__.py)."""
import hashlib
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BENCH = os.path.join(ROOT, "bench.py")      # child-process modes re-enter through bench.py

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

# Integer-VALU roofline of the Blake2s kernel: one compression = 977 VALU lane-ops in the compiled kernel (v_add3_u32 / v_xor_b32 /
# v_alignbit_b32; llvm-objdump of k_merkle_layer), and the chip retires 256 CU x 4 SIMD x 16 int lanes/clk x 2.4 GHz = 39.3 T such
# lane-ops/s (half the fp32-FMA issue rate; tools/ubench_blake.hip measures 39.9 G compressions/s = 39.0 T lane-ops/s in registers).
VALU_OPS_PER_COMPRESSION = 977
VALU_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12

FIB19 = "+++++++++++++++++>+>+<<[->>[->+>+<<]<[->>+<<]>>[-<+>]>[-<<<+>>>]<<<<]>>."  # tests/golden/programs/fib19.bf (workload input)

# Synthetic padded traces: "+"*a "[>" "+"*b "[>+<-]<-]" — the Memory component lands exactly on
# 2^k domain rows for (a, b) = (14, 250 * 2^(k-20)); proved with LOG_MAX_ROWS = k.
SWEEP = {k: (14, 250 << (k - 20)) for k in range(20, 27)}


def sweep_program(k):
    a, b = SWEEP[k]
    return "+" * a + "[>" + "+" * b + "[>+<-]<-]"


def load_package():
    name = "stwo_brainfuck_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "stwo-brainfuck_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def kernel_sources_sha256():
    """SHA-256 over the sources of the dominant kernel (what a committed counter file must have been measured on)."""
    h = hashlib.sha256()
    for rel in ("stwo-brainfuck_amd/csrc/merkle.hip", "stwo-brainfuck_amd/csrc/kernels.h", "stwo-brainfuck_amd/csrc/m31.h"):
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()


def committed_digests():
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "fib19_lmr24_oracle_proof.json")))
    except Exception:
        return {}


MAIN_COLS = [8, 8, 4, 9, 13, 13, 11, 11, 11, 11, 11, 11, 7]      # TraceColumn::count().0 per component, claim order (mod.rs:85-99)
LOGUP_COLS = [1, 1, 1, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1]


def pick_device(local_rank, n_visible, override=None):
    """One process per GPU: rank r drives device LOCAL_RANK. A launcher that narrows each rank's view to its own GPU
    (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank) leaves one visible device, numbered 0, on every rank."""
    if override is not None:
        return override
    return local_rank if local_rank < n_visible else local_rank % max(n_visible, 1)


def want_digest(conv, log_max_rows):
    """The committed digest of the CPU oracle's proof of the bench workload under these conventions (tests/golden), or None."""
    return next((d for d in committed_digests().values() if tuple(d.get("conventions", ())) == tuple(conv) and d.get("log_max_rows") == log_max_rows), None)
