"""CPU suite: the self-launcher of bench.py (`python3 bench.py --gpus N` without torch.distributed.run): the environment each rank gets, the
relay of rank 0's one JSON line, a failing rank -> non-zero exit with the others ended, the time limit, and --gpus / WORLD_SIZE disagreement."""
import importlib.util
import io
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


FAKE_RANK = r"""
import json, os, sys, time
out_dir, mode = sys.argv[1], sys.argv[2]
rank = int(os.environ["RANK"])
json.dump({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY", "BFHIP_BENCH_SELF_LAUNCHED")},
          open(os.path.join(out_dir, f"env{rank}.json"), "w"))
if mode == "ok":
    print("noise from rank", rank)
    if rank == 0:
        print(json.dumps({"metric": "m", "value": 1.5, "n_gpus": int(os.environ["WORLD_SIZE"])}))
elif mode == "fail":
    if rank == 1:
        sys.exit(7)
    time.sleep(120)
elif mode == "hang":
    time.sleep(120)
elif mode == "silent":
    pass
"""


def run_fake(tmp_path, mode, n=3, timeout=60):
    bench = bench_module()
    script = tmp_path / "fake_rank.py"
    script.write_text(FAKE_RANK)
    buf = io.StringIO()
    t0 = time.time()
    rc = bench.launch_ranks([sys.executable, str(script), str(tmp_path), mode], n, timeout, out=buf, poll=0.05)
    return rc, buf.getvalue(), time.time() - t0


def test_rank_environments():
    envs = bench_module().rank_environments(4, 29999, {"PATH": "/bin", "WORLD_SIZE": "stale"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29999" and e["PATH"] == "/bin" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)       # dmabuf IPC: RCCL across processes needs it on this pool


def test_launcher_starts_n_ranks_and_relays_rank0s_line(tmp_path):
    rc, text, _ = run_fake(tmp_path, "ok", n=3)
    assert rc == 0
    lines = [l for l in text.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "m", "value": 1.5, "n_gpus": 3}      # ONE line: the other ranks' output is not relayed
    ports = set()
    for r in range(3):
        env = json.load(open(tmp_path / f"env{r}.json"))
        assert env["RANK"] == str(r) and env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "3" and env["MASTER_ADDR"] == "127.0.0.1"
        assert env["BFHIP_BENCH_SELF_LAUNCHED"] == "1"
        ports.add(env["MASTER_PORT"])
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536


def test_a_failing_rank_ends_the_others_and_the_exit_code_is_non_zero(tmp_path):
    rc, text, took = run_fake(tmp_path, "fail", n=3)
    assert rc == 1 and text.strip() == ""
    assert took < 30, took                      # rank 0 and rank 2 would have slept 120 s


def test_the_time_limit_ends_hung_ranks(tmp_path):
    rc, text, took = run_fake(tmp_path, "hang", n=2, timeout=2)
    assert rc == 124 and took < 30


def test_a_missing_line_is_a_failure(tmp_path):
    rc, text, _ = run_fake(tmp_path, "silent", n=2)
    assert rc == 1 and text.strip() == ""


def test_gpus_and_world_size_must_agree():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "must agree" in r.stderr and "WORLD_SIZE=3" in r.stderr, r.stderr


def test_launcherless_run_without_gpus_fails_loudly():
    """On a box without GPUs `python3 bench.py --gpus 2` starts two ranks, both fail (no CPU fallback), and the launcher exits non-zero without a line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: covered by the -m gpu launcher test")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--no-cpu-baseline", "--launch-timeout", "200"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0, r.stdout
    assert not any(l.startswith("{") for l in r.stdout.splitlines())
