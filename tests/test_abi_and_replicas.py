"""CPU: the C-ABI library loads and exports every symbol include/bfhip.h declares (no compute calls without a GPU), fails loudly
without a GPU, and the N>1 "replicas" timing protocol works over gloo with world_size 2."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT, TESTHOOKS_LIBRARY


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


def test_default_library_has_no_test_hooks():
    """The release build must not contain the test hooks: no lookup of a mock's symbol, no environment variable that makes it
    dlopen an arbitrary path, no host delay for the mailboxes. The hooks build (same sources, -DBFHIP_TEST_HOOKS in api.hip / comm.hip) has them and
    exports the same entry points."""
    pkg_dir = os.path.join(ROOT, "stwo-brainfuck_amd")
    blob = open(os.path.join(pkg_dir, "libbfhip.so"), "rb").read()
    for needle in (b"bfhip_mock_self_copy", b"BFHIP_RCCL_LIBRARY", b"BFHIP_MAILBOX_TEST_DELAY_MS"):
        assert needle not in blob, f"the default libbfhip.so contains the test hook {needle!r}"
    assert b"mock" not in blob.lower().replace(b"mockingbird", b""), "the default libbfhip.so mentions a mock"
    hooks = open(TESTHOOKS_LIBRARY, "rb").read()
    for needle in (b"bfhip_mock_self_copy", b"BFHIP_RCCL_LIBRARY", b"BFHIP_MAILBOX_TEST_DELAY_MS"):
        assert needle in hooks
    H = ctypes.CDLL(TESTHOOKS_LIBRARY)
    for s in declared_symbols():
        assert hasattr(H, s), f"libbfhip_testhooks.so does not export {s}"


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
# a rank 0 that cannot create the id still takes part in the broadcast, and EVERY rank gets an exception (no rank is left waiting)
def broken():
    raise RuntimeError("no librccl here")
try:
    rep.share_unique_id(dist, broken)
    raise SystemExit("share_unique_id should have raised on rank %d" % rank)
except RuntimeError as e:
    assert ("no librccl here" in str(e)) if rank == 0 else ("all-zero marker" in str(e)), str(e)
dist.barrier()
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
            "ctx_create_null_out": lambda: L.bfhip_ctx_create(0, 20, None),
            "pool_create_null_out": lambda: L.bfhip_pool_create(0, 2, 20, None),
            "pool_create_zero": lambda: L.bfhip_pool_create(0, 0, 20, ctypes.byref(ctypes.c_void_p())),
            "pool_destroy_null": lambda: L.bfhip_pool_destroy(None),
            "pool_batch_null": lambda: L.bfhip_prove_batch(None, None, 0, 20, None, None, None, None),
            "pool_programs_null": lambda: L.bfhip_prove_batch_brainfuck(None, None, None, None, 0, 20, None, None, None, None),
            "pool_ctx_null": lambda: L.bfhip_pool_ctx(None, 0, None),
        }
        print(calls[sys.argv[1]]())
    """) % os.path.join(root, "tests")
    expect = {"compile_null": -1, "run_null_code": -1, "table_bad_component": -1, "verify_null": 1, "verify_no_error_buffer": 1, "ctx_create_null_out": -1,
              "pool_create_null_out": -1, "pool_create_zero": -1, "pool_destroy_null": 0, "pool_batch_null": -1, "pool_programs_null": -1, "pool_ctx_null": -1}
    for name in ["compile_null", "run_null_code", "run_null_counts", "table_null", "table_bad_component", "verify_null", "verify_no_error_buffer",
                 "ctx_destroy_null", "trace_destroy_null", "free_host_null", "component_shape_null", "ctx_create_null_out", "pool_create_null_out",
                 "pool_create_zero", "pool_destroy_null", "pool_batch_null", "pool_programs_null", "pool_ctx_null"]:
        r = subprocess.run([sys.executable, "-c", prog, name], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (name, r.returncode, r.stderr[-300:])
        if name in expect:
            assert int(r.stdout.strip().splitlines()[-1]) == expect[name], (name, r.stdout)


def test_rccl_exchange_bookkeeping_through_a_mock_library(tmp_path):
    """RcclComm::exchange (comm.hip) driven through a test double of the 10 RCCL entry points (tests/mock_rccl.c, ranks = threads, host
    buffers): real ncclSend/ncclRecv need one GPU per rank, which no test box has. Checks what the transport itself is responsible for —
    blocks to oneself (copied, matched in order), zero-byte blocks (skipped on both sides), several blocks per peer (order kept), the
    counters — and that a size mismatch between a send and its receive is an ERROR on the ranks involved, not a hang."""
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mock = tmp_path / "libmock_rccl.so"
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", str(mock), os.path.join(root, "tests", "mock_rccl.c"), "-lpthread"])
    prog = textwrap.dedent("""
        import ctypes, sys, threading
        import numpy as np
        sys.path.insert(0, %r)
        from conftest import load_package
        L = load_package().lib()
        N = 4
        uid = (ctypes.c_uint8 * 128)()
        assert L.bfhip_rccl_unique_id(uid) == 0

        def blocks(src, dst, bad):
            # what rank src sends to rank dst, in order: (words, fill); a zero-byte block sits in the middle
            out = [(5 + src, 100 * src + dst), (0, 0), (3, 1000 + 10 * src + dst), (7 + dst, 7)]
            if bad and (src, dst) == (1, 2):
                out[2] = (4, out[2][1])            # rank 1 announces 4 words where rank 2 expects 3
            return out

        def run(rank, bad, res):
            sends, recvs, keep, want = [], [], [], []
            for peer in range(N):
                for words, fill in blocks(rank, peer, bad):
                    a = np.full(words, fill, dtype=np.uint32); keep.append(a)
                    sends.append((peer, a))
                for words, fill in blocks(peer, rank, False):
                    a = np.zeros(words, dtype=np.uint32); keep.append(a)
                    recvs.append((peer, a)); want.append(np.full(words, fill, dtype=np.uint32))
            def pack(lst):
                n = len(lst)
                return (n, (ctypes.c_uint32 * n)(*[p for p, _ in lst]), (ctypes.c_void_p * n)(*[a.ctypes.data for _, a in lst]), (ctypes.c_size_t * n)(*[a.nbytes for _, a in lst]))
            stats = (ctypes.c_uint64 * 4)()
            rc = L.bfhip_rccl_exchange_raw(uid, rank, N, *pack(sends), *pack(recvs), stats)
            ok = rc == 0 and all(np.array_equal(a, w) for (_, a), w in zip(recvs, want))
            sent = sum(a.nbytes for p, a in sends if p != rank)
            res[rank] = (rc, ok, list(stats), sent, L.bfhip_last_error().decode() if rc else "")

        for bad in (False, True):
            res = [None] * N
            th = [threading.Thread(target=run, args=(r, bad, res)) for r in range(N)]
            [t.start() for t in th]; [t.join(60) for t in th]
            assert not any(t.is_alive() for t in th), "a rank hangs"
            if not bad:
                for r in range(N):
                    rc, ok, stats, sent, err = res[r]
                    assert rc == 0 and ok, (r, res[r])
                    assert stats == [0, 0, 1, sent], (r, stats, sent)
            else:
                assert res[2][0] != 0 and "RCCL" in res[2][4] or "invalid usage" in res[2][4], res[2]      # the receiver of the wrong size
                assert res[0][0] == 0 and res[0][1]                                                          # an uninvolved rank is unaffected
            if not bad:
                uid = (ctypes.c_uint8 * 128)(); assert L.bfhip_rccl_unique_id(uid) == 0     # a fresh group for the failing round
        print("ok")
    """) % os.path.join(root, "tests")
    # BFHIP_RCCL_LIBRARY is a test hook: only libbfhip_testhooks.so (-DBFHIP_TEST_HOOKS) reads it
    env = dict(os.environ, BFHIP_RCCL_LIBRARY=str(mock), BFHIP_LIBRARY=TESTHOOKS_LIBRARY)
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-1500:] + r.stderr[-1500:]


# ---- bench.py host logic (no GPU): the strong-scaling digest, the SimdBackend-shaped work counts, the counter-file guard -------------------
def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_cpu_budget_reports_effective_cores():
    """cpu_baseline.cores must be the cores the process can actually use: min(threads in the affinity mask, cgroup CPU quota) — not the size of
    an OpenMP team that time-shares them."""
    b = _bench()
    bud = b.host_cpu_budget()
    assert bud["affinity_threads"] >= 1 and 1 <= bud["cores_effective"] <= bud["affinity_threads"]
    if bud["quota_cores"] is not None:
        assert bud["cores_effective"] <= max(1, int(bud["quota_cores"]))


def test_bench_simdbackend_work_counts_fib19():
    """What a SimdBackend-shaped prover hashes and transforms for fib19: more than the HIP path's own compression count (it does not know the
    16x replication: 674 M there), and the butterflies of 213 full-size columns."""
    b = _bench()
    comps, bflies = b.simdbackend_work_counts([24, 22, 11, 22, 19, 11, 4, 20, 19, 4, 20, 20, 4], 24)
    assert 700e6 < comps < 800e6 and 1.5e10 < bflies < 3e10
    # a single 2^4 component: the tree of its 8 + 4 columns and the 21 IsFirst columns are small numbers one can count by hand
    c2, b2 = b.simdbackend_work_counts([4] * 13, 4)
    assert c2 > 0 and b2 > 0 and c2 < 5000


def test_bench_kernel_source_digest_is_stable_and_sensitive(tmp_path):
    b = _bench()
    h = b.kernel_sources_sha256()
    assert h == b.kernel_sources_sha256() and len(h) == 64


def test_bench_roofline_from_the_committed_rocprof_summary():
    """roofline.frac_rocprof: recomputed from profiles/rNN_roofline_single_stream_kernel_stats.csv while that run's kernel sources are the current ones; a stale
    summary yields null WITH the reason (never a number measured on another kernel). Either way the shape is fixed."""
    b = _bench()
    out = b.roofline_from_committed_rocprof(669777920, 78.0)
    assert set(out) >= {"frac_rocprof", "frac_rocprof_source", "frac_range_this_round"} and isinstance(out["frac_rocprof_source"], str)
    if out["frac_rocprof"] is None:
        assert "STALE" in out["frac_rocprof_source"] or "no committed" in out["frac_rocprof_source"] or ":" in out["frac_rocprof_source"]
    else:
        assert 0.5 < out["frac_rocprof"] < 1.0 and out["avg_launch_us_rocprof"] > 0
        # the arithmetic a reader would do by hand: compressions x 977 lane-ops / (average launch x launches per proof) / (256 CU x 4 SIMD x 16 lanes x 2.4 GHz)
        want = 669777920 * 977 / (out["avg_launch_us_rocprof"] * 1e-6 * 78.0) / (256 * 4 * 16 * 2.4e9)
        assert abs(want - out["frac_rocprof"]) < 2e-3
    assert b.roofline_from_committed_rocprof(None, 78.0)["frac_rocprof"] is None
