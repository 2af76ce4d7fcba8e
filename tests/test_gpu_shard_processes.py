"""-m gpu: a shard group whose ranks are PROCESSES (one libbfhip context each) — the deployment's control flow: the 128-byte unique id travels
over torch.distributed (gloo), every process joins with bfhip_ctx_join_rccl_group, RcclComm moves DEVICE buffers, the waits are bounded across
processes. The box has one GPU and real librccl refuses two ranks on one device, so the RCCL entry points come from the process-per-rank test
double tests/mock_rccl_ipc.cpp (BFHIP_RCCL_LIBRARY; POSIX shared-memory rendezvous, hipIpc memory handles). Still unexercised afterwards:
librccl itself and xGMI."""
import glob
import hashlib
import json
import os
import signal
import socket
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROGS = os.path.join(ROOT, "tests", "golden", "programs")
# BFHIP_RCCL_LIBRARY (the RCCL double) is a test hook: only the -DBFHIP_TEST_HOOKS build of the library reads it (csrc/Makefile)
TESTHOOKS_LIBRARY = os.path.join(ROOT, "stwo-brainfuck_amd", "libbfhip_testhooks.so")


def build_ipc_double():
    """tests/libmock_rccl_ipc.so (also built by __graft_entry__.build()); rebuilt when the source is newer."""
    src, so = os.path.join(ROOT, "tests", "mock_rccl_ipc.cpp"), os.path.join(ROOT, "tests", "libmock_rccl_ipc.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", so, src,
                               "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"])
    return so


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def start_ranks(tmp_path, world, program, inp, lmr, proofs=1, extra_env=None, conv=(0, 0, 0, 0), policy=-1):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   BFHIP_RCCL_LIBRARY=build_ipc_double(), BFHIP_LIBRARY=TESTHOOKS_LIBRARY, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        out = str(tmp_path / f"rank{r}.json")
        log = open(str(tmp_path / f"rank{r}.log"), "w")
        p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), "--program", os.path.join(PROGS, program), "--input-hex", inp.hex(),
                              "--log-max-rows", str(lmr), "--out", out, "--proofs", str(proofs), "--conventions", ",".join(str(v) for v in conv), "--shard-policy", str(policy)],
                             env=env, stdout=log, stderr=subprocess.STDOUT)
        procs.append((p, out, log))
    return procs


def finish(procs, timeout):
    """Waits for every rank (bounded); kills exactly the PIDs started here if the limit passes. Returns [(exit code, result dict, log text)]."""
    t_end = time.time() + timeout
    res = []
    for p, out, log in procs:
        try:
            p.wait(timeout=max(1.0, t_end - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
        log.close()
        try:
            data = json.load(open(out))
        except Exception:
            data = {}
        res.append((p.returncode, data, open(log.name).read()[-3000:]))
    for f in glob.glob("/dev/shm/bfhip_mock_*"):      # a killed rank cannot unlink the rendezvous segment
        try:
            os.remove(f)
        except OSError:
            pass
    return res


_ORACLE_COLLATZ = {}


@pytest.mark.with_poseidon
@pytest.mark.parametrize("world,policy", [(2, 0), (2, -1), (4, -1), (4, 1)])
def test_process_group_proves_collatz_like_one_gpu(pkg, oracle, conv, tmp_path, world, policy):
    """policy: bfhip_ctx_set_shard_policy. -1 = automatic: a group of TWO ranks of an RCCL transport replicates the transforms (no column -> row exchange over its single
    link), larger groups exchange; 0 / 1 force either."""
    code, inp, lmr = open(os.path.join(PROGS, "collatz.bf")).read(), b"7\n", 21
    c1 = pkg.Context(0, max_log_domain=lmr + 2)
    try:
        single = pkg.prove_brainfuck(code, inp, ctx=c1, log_max_rows=lmr)
    finally:
        c1.close()
    if conv not in _ORACLE_COLLATZ:             # the CPU oracle's Poseidon252 proof takes a minute: once per convention set, not once per (world, policy)
        _ORACLE_COLLATZ[conv] = oracle.prove(code, inp, log_max_rows=lmr)[0]
    assert single == _ORACLE_COLLATZ[conv]
    res = finish(start_ranks(tmp_path, world, "collatz.bf", inp, lmr, proofs=2, conv=conv, policy=policy), timeout=600)
    replicated = policy == 1 or (policy == -1 and world == 2)
    for r, (rc, data, log) in enumerate(res):
        assert rc == 0, (r, rc, data, log)
        assert "BFHIP_RCCL_LIBRARY" in data["transport"], data
        assert data["proofs"] == [hashlib.sha256(single).hexdigest()] * 2, (r, data)
        assert open(str(tmp_path / f"rank{r}.json") + ".proof", "rb").read() == single, f"rank {r}: bytes differ from the single-GPU proof"
        st = data["group_stats"]
        assert st["all_gathers"] >= 4 and st["max_reduces"] >= 2 and st["bytes_sent"] > 0, st
        assert (st["exchanges"] == 0) if replicated else (st["exchanges"] >= 2), (policy, st)


@pytest.mark.single_conv
@pytest.mark.parametrize("world", [2, 4])
def test_process_group_proves_fib19_like_one_gpu(pkg, tmp_path, world):
    """BASELINE config 2 / 4 size (2^24 domain rows) over `world` processes; the digest is the committed one of the CPU oracle's proof."""
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "fib19_lmr24_oracle_proof.json")))["stwo"]
    res = finish(start_ranks(tmp_path, world, "fib19.bf", b"", 24), timeout=900)
    for r, (rc, data, log) in enumerate(res):
        assert rc == 0, (r, rc, data, log)
        assert data["proofs"] == [want["sha256"]], (r, data)
        assert os.path.getsize(str(tmp_path / f"rank{r}.json") + ".proof") == want["proof_bytes"]


def test_a_killed_rank_fails_the_others_within_the_timeout(tmp_path):
    """Rank 1 is killed (SIGKILL, the exact PID) while the group is proving in a loop: rank 0 must come back with an ERROR from the library
    within BFHIP_COMM_TIMEOUT_S — not hang, and not exit as if it had proved."""
    limit = 20
    procs = start_ranks(tmp_path, 2, "collatz.bf", b"7\n", 21, proofs=100000, extra_env={"BFHIP_COMM_TIMEOUT_S": str(limit)})
    out0 = procs[0][1]
    t0 = time.time()
    while time.time() - t0 < 300:                      # until the group has produced a few proofs
        try:
            if len(json.load(open(out0)).get("proofs", [])) >= 3:
                break
        except Exception:
            pass
        if procs[0][0].poll() is not None:
            break
        time.sleep(0.05)
    else:
        finish(procs, 1)
        pytest.fail("the group never produced a proof")
    victim = procs[1][0]
    os.kill(victim.pid, signal.SIGKILL)
    t_kill = time.time()
    try:
        procs[0][0].wait(timeout=limit + 30)
    except subprocess.TimeoutExpired:
        finish(procs, 1)
        pytest.fail(f"rank 0 still running {limit + 30} s after its peer was killed")
    took = time.time() - t_kill
    res = finish(procs, 5)
    rc0, data0, log0 = res[0]
    assert rc0 == 3, (rc0, data0, log0)               # BfhipError caught by the worker: the library reported the failure
    assert "error" in data0 and len(data0["proofs"]) >= 3, data0
    assert took <= limit + 10, took


def test_bench_launches_its_own_ranks_and_headlines_the_shard_group(tmp_path):
    """`python3 bench.py --gpus 2` with no launcher: bench.py starts the two ranks itself (fresh processes), they form a shard group (here both on
    device 0 through the process-per-rank double, torch.distributed over gloo), and rank 0's ONE line reports n_gpus 2, scaling "strong", the
    group's proof with the SHA-256 of the one-GPU proof, the replicas beside it."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BFHIP_RCCL_LIBRARY=build_ipc_double(), BFHIP_LIBRARY=TESTHOOKS_LIBRARY, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--device", "0", "--steps", "3", "--warmup", "1",
                        "--no-extra-stages", "--group-inflight", "2", "--no-local-probe", "--no-cpu-baseline", "--launch-timeout", "900"], env=env, capture_output=True, text=True, timeout=1000)
    for f in glob.glob("/dev/shm/bfhip_mock_*"):
        os.remove(f)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "fib19_lmr24_oracle_proof.json")))["stwo"]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 3
    assert line["parity_checked"] is True and line["parity"]["proof_sha256"] == want["sha256"]
    head = line["strong_scaling"]["workloads"]["fib19"]
    assert head["identical_to_n1"] is True and head["proof_sha256"] == want["sha256"] and head["speedup_vs_n1"] == line["speedup_vs_n1"]
    assert line["replicas"]["scaling"] == "weak" and line["replicas"]["value"] > 0
    assert line["config"]["ranks_started_by"].startswith("bench.py itself")
    assert abs(line["value"] - line["config"]["cells_per_proof"] / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6      # one proof's cells, not N times
    # r06: the one-GPU reference is rank 0 ALONE (the others idle at a barrier), the replicas' slowest-rank time rides beside it; per-collective latency
    # histogram of the timed proofs (count, GPU-side and host-side p50 / p90 / max per kind) next to the bytes
    assert "rank 0 alone" in head["n1_what"] and line["roofline"]["frac_rocprof"] is None and "share" in line["roofline"]["frac_rocprof_source"]
    lat = head["collective_latency_us_rank0"]
    assert lat["all_gather"]["count"] >= 3 * 4 and lat["all_gather"]["gpu_us"]["p50"] > 0 and lat["all_gather"]["host_us"]["max"] > 0, lat
    # two ranks of an RCCL transport: the automatic shard policy replicates the transforms — no column -> row exchange over the pair's single link
    assert head["shard_policy"] == {"requested": -1, "replicated_transforms": True} and lat["exchange"]["count"] == 0 and head["per_proof_rank0"]["exchanges"] == 0, (head["shard_policy"], lat)
    # a pool of 3 per rank at the metric's size, all ranks at once: the node's deployable throughput (weak scaling; here both pools share ONE GPU)
    rp = line["replicas_pool"]
    assert "error" not in rp and rp["value"] > 0 and rp["in_flight_per_gpu"] == 3 and len(rp["proof_sha256"]) == 64 and abs(rp["value"] - 2 * rp["cells_per_s_per_gpu"]) / rp["value"] < 1e-3, rp
    # --group-inflight 2: two shard groups proving at the same time (two contexts, host threads and communicators per rank), every proof the one-GPU bytes
    two = line["strong_scaling"]["workloads"]["fib19_2_in_flight"]
    assert two["in_flight"] == 2 and two["identical_to_the_headline_proof"] is True and two["proofs_timed"] == 6 and two["ms_per_proof"] > 0 and "error" not in two, two
    assert abs(two["gain_vs_one_in_flight"] - head["ms_per_proof"] / two["ms_per_proof"]) < 0.01


def test_bench_prints_the_replicas_line_when_the_group_never_comes_back():
    """--group-timeout: a shard group that does not finish in time (here: a limit of 0 s) cannot be interrupted — rank 0 prints the replicas line it measured
    before the group formed (contract protocol, K steps), flagged with shard_group_error, and every rank leaves with exit code 3 (a process stuck in a collective did
    not succeed: ADVICE r05) — the launcher still relays rank 0's line: a multi-GPU run always yields a line, and a non-zero code when something hung."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BFHIP_RCCL_LIBRARY=build_ipc_double(), BFHIP_LIBRARY=TESTHOOKS_LIBRARY, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--device", "0", "--steps", "3", "--warmup", "1",
                        "--no-extra-stages", "--no-local-probe", "--no-cpu-baseline", "--group-timeout", "0", "--launch-timeout", "600"], env=env, capture_output=True, text=True, timeout=700)
    for f in glob.glob("/dev/shm/bfhip_mock_*"):
        os.remove(f)
    assert r.returncode != 0 and "exit code 3" in r.stderr, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 3 and "did not finish within 0 s" in line["shard_group_error"]
    assert line["parity_checked"] is True and line["value"] == line["replicas"]["value"] > 0
    assert abs(line["value"] - 2 * line["config"]["cells_per_proof"] / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6      # two proofs per step: one per rank


def test_bench_replicas_headline_and_python_threads_in_flight():
    """The two other shapes of the line: `--gpus 2 --replicas` (N independent proofs, weak scaling, no group is formed) and `--inflight 2` at N = 1 (two contexts and host
    threads per step: the round-5 pattern, kept for comparison with the pool)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--replicas", "--dist-backend", "gloo", "--device", "0", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--launch-timeout", "600"], env=env, capture_output=True, text=True, timeout=700)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["parity_checked"] is True and "strong_scaling" not in line and "shard_group_error" not in line
    assert abs(line["value"] - 2 * line["config"]["cells_per_proof"] / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6 and line["config"]["parallelism"] == "replicas"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--inflight", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-sweep"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert line["n_gpus"] == 1 and line["config"]["proofs_in_flight_per_gpu"] == 2 and line["roofline"] is None and line["parity_checked"] is True
    assert abs(line["value"] - 2 * line["config"]["cells_per_proof"] / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "must agree" in r.stderr


def test_bench_falls_back_to_the_replicas_when_the_group_cannot_be_formed(tmp_path):
    """The RCCL entry points cannot be loaded (BFHIP_RCCL_LIBRARY names a file that does not exist): joining fails on every rank, the ranks agree on that over
    torch.distributed, and the line is the replicas' (weak scaling) with shard_group_error saying why — exit code 0, one line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BFHIP_RCCL_LIBRARY=str(tmp_path / "no_such_librccl.so"), BFHIP_LIBRARY=TESTHOOKS_LIBRARY)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--device", "0", "--steps", "2", "--warmup", "1",
                        "--no-extra-stages", "--no-local-probe", "--no-cpu-baseline", "--launch-timeout", "600"], env=env, capture_output=True, text=True, timeout=700)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "joining the shard group failed" in line["shard_group_error"] and "RCCL" in line["shard_group_error"]
    assert line["parity_checked"] is True and line["roofline"]["kernel"] == "k_merkle_layer"
    assert abs(line["value"] - 2 * line["config"]["cells_per_proof"] / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6
