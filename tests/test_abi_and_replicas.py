"""CPU: the C-ABI library loads and exports every symbol include/bfhip.h declares (no compute calls without a GPU), fails loudly
without a GPU, and the N>1 "replicas" timing protocol works over gloo with world_size 2."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "bfhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(bfhip_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"libbfhip.so does not export {s}"


def test_header_cites_reference_lines():
    hdr = open(os.path.join(ROOT, "include", "bfhip.h")).read()
    assert hdr.count("mod.rs:") >= 8   # every entry point names the reference interface it replaces


def test_no_cpu_fallback_without_gpu(pkg):
    if pkg.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(pkg.BfhipError, match="no CPU fallback|no HIP device"):
        pkg.Context(0, 22)
    with pytest.raises(pkg.BfhipError):
        pkg.prove_brainfuck("+", b"", log_max_rows=12)


def test_product_does_not_reference_the_oracle():
    """The shipped path must never import, link or load anything under oracle/."""
    pkg_dir = os.path.join(ROOT, "stwo-brainfuck_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle/" not in txt and "libbforacle" not in txt and "orc_" not in txt, f"{f} references the oracle"
    out = subprocess.run(["ldd", os.path.join(pkg_dir, "libbfhip.so")], capture_output=True, text=True).stdout
    assert "bforacle" not in out


WORKER = r"""
import os, sys, time, importlib.util
import torch.distributed as dist
root = sys.argv[1]
spec = importlib.util.spec_from_file_location("replicas", os.path.join(root, "stwo-brainfuck_amd", "replicas.py"))
rep = importlib.util.module_from_spec(spec); spec.loader.exec_module(rep)
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank = dist.get_rank()
calls = []
def step():
    calls.append(1); time.sleep(0.02 * (rank + 1)); return ("proof", {"total": 0.0})
dt, last = rep.timed_region(step, steps=3, warmup=1, dist=dist)
cells = rep.aggregate_units(1000 + rank, dist=dist)
assert len(calls) == 4 and last[0] == "proof"
assert dt >= 0.02 * 2 * 3 * 0.9, dt          # MAX over ranks: the slow rank (rank 1) sets the time
assert cells == 2001
assert rep.rank_device(rank, 8) == rank
# control plane of a shard group (one proof over several GPUs): rank 0's 128-byte unique id reaches every rank over gloo
made = []
uid = rep.share_unique_id(dist, lambda: (made.append(1), bytes(range(128)))[1])
assert uid == bytes(range(128)), uid
assert len(made) == (1 if rank == 0 else 0)
print("ok", rank, round(dt, 3))
dist.destroy_process_group()
"""


def test_replicas_protocol_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29517",
                        str(script), ROOT], capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_rust_ffi_is_in_step_with_the_header():
    """bindings/rust/bfhip_sys.rs is generated from include/bfhip.h (tools/gen_rust_ffi.py): regenerating must reproduce the committed
    file, and it must declare every function the header declares."""
    import re
    committed = open(os.path.join(ROOT, "bindings", "rust", "bfhip_sys.rs")).read()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(os.path.join(ROOT, "bindings", "rust", "bfhip_sys.rs")).read() == committed, "bfhip_sys.rs is stale: run tools/gen_rust_ffi.py"
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "bfhip.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(bfhip_\w+)\s*\(", header))
    in_rust = set(re.findall(r"pub fn (bfhip_\w+)\(", committed))
    assert declared == in_rust, declared ^ in_rust


def test_hip_backend_covers_the_trait_surface():
    """bindings/rust/hip_backend.rs (source only: no Rust toolchain in the image) implements every stwo backend trait the reference's
    generic parameter needs, every method INTEGRATION.md section 2 maps, and only calls FFI functions that
    bfhip_sys.rs declares."""
    import re
    src = open(os.path.join(ROOT, "bindings", "rust", "hip_backend.rs")).read()
    for tr in ["Backend for HipBackend", "BackendForChannel<Blake2sMerkleChannel> for HipBackend", "ColumnOps<BaseField> for HipBackend", "ColumnOps<SecureField> for HipBackend",
               "ColumnOps<Blake2sHash> for HipBackend", "FieldOps<BaseField> for HipBackend", "FieldOps<SecureField> for HipBackend", "PolyOps for HipBackend",
               "MerkleOps<Blake2sMerkleHasher> for HipBackend", "QuotientOps for HipBackend", "FriOps for HipBackend", "AccumulationOps for HipBackend",
               "GrindOps<Blake2sChannel> for HipBackend", "GkrOps for HipBackend", "ComponentProver<HipBackend> for FrameworkComponent<E>"]:
        assert f"impl {tr}" in src or f"impl<E: BrainfuckEval> {tr}" in src, f"missing impl {tr}"
    for method in ["bit_reverse_column", "batch_inverse", "precompute_twiddles", "interpolate_columns", "evaluate_polynomials", "eval_at_point", "extend", "commit_on_layer",
                   "accumulate_quotients", "fold_line", "fold_circle_into_line", "accumulate", "generate_secure_powers", "grind", "evaluate_constraint_quotients_on_domain",
                   "zeros", "to_cpu", "from_iter"]:
        assert re.search(rf"fn {method}\b", src), f"missing method {method}"
    declared = set(re.findall(r"pub fn (bfhip_\w+)\(", open(os.path.join(ROOT, "bindings", "rust", "bfhip_sys.rs")).read()))
    used = set(re.findall(r"sys::(bfhip_\w+)", src))
    assert used and used <= declared, used - declared
    integration = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for entry in used:
        assert entry in integration, f"{entry} is used by hip_backend.rs but not mapped in INTEGRATION.md"


def test_host_entries_survive_null_and_invalid_arguments():
    """Every host-only entry point called with null pointers, zero sizes or an unknown component in a child process each (a crash would be a
    signal exit code): all return, with -1 / 1 where the reference would panic."""
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = textwrap.dedent("""
        import ctypes, sys
        sys.path.insert(0, %r)
        from conftest import load_package
        L = load_package().lib()
        n, m = ctypes.c_size_t(), ctypes.c_size_t()
        z = ctypes.c_size_t(0)
        calls = {
            "compile_null": lambda: L.bfhip_host_compile(None, None, z, ctypes.byref(n)),
            "run_null_code": lambda: L.bfhip_host_run(None, None, z, None, z, ctypes.byref(n), None, z, ctypes.byref(m)),
            "run_null_counts": lambda: L.bfhip_host_run(b"+", None, z, None, z, None, None, z, None),
            "table_null": lambda: L.bfhip_host_table(None, z, None, z, 0, None, z, ctypes.byref(n), ctypes.byref(m)),
            "table_bad_component": lambda: L.bfhip_host_table((ctypes.c_uint32 * 7)(), ctypes.c_size_t(1), (ctypes.c_uint32 * 1)(43), ctypes.c_size_t(1), 99, None, z, ctypes.byref(n), ctypes.byref(m)),
            "verify_null": lambda: L.bfhip_verify_brainfuck(None, z, 20, None, z),
            "verify_no_error_buffer": lambda: L.bfhip_verify_brainfuck(b"{}", ctypes.c_size_t(2), 20, None, z),
            "ctx_destroy_null": lambda: L.bfhip_ctx_destroy(None),
            "trace_destroy_null": lambda: L.bfhip_trace_destroy(None, None),
            "free_host_null": lambda: L.bfhip_free_host(None),
            "component_shape_null": lambda: L.bfhip_component_shape(0, None, None, None),
        }
        print(calls[sys.argv[1]]())
    """) % os.path.join(root, "tests")
    expect = {"compile_null": -1, "run_null_code": -1, "table_bad_component": -1, "verify_null": 1, "verify_no_error_buffer": 1}
    for name in ["compile_null", "run_null_code", "run_null_counts", "table_null", "table_bad_component", "verify_null", "verify_no_error_buffer",
                 "ctx_destroy_null", "trace_destroy_null", "free_host_null", "component_shape_null"]:
        r = subprocess.run([sys.executable, "-c", prog, name], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (name, r.returncode, r.stderr[-300:])
        if name in expect:
            assert int(r.stdout.strip().splitlines()[-1]) == expect[name], (name, r.stdout)
