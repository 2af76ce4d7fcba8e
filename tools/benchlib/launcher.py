"""`python3 bench.py --gpus N` without a launcher, and host-thread pinning (see tools/benchlib/__init__.py)."""
import os
import sys
import time


def gpu_local_cpus(device):
    """The cores next to GPU `device` (sysfs local_cpulist of its PCI function), or None when that cannot be read."""
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) != 0:
            return None
        text = open(f"/sys/bus/pci/devices/{buf.value.decode().lower()}/local_cpulist").read().strip()
        cpus = set()
        for part in text.split(","):
            if "-" in part:
                a, b = part.split("-"); cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        return cpus or None
    except Exception:
        return None


class pinned_host_thread:
    """The proving thread on ONE core next to its GPU for the duration of a timed region (restored afterwards: the CPU baseline and child
    processes use every core). A proof is ~10 Fiat-Shamir round trips with the GPU idle in each; a thread that the scheduler migrates while it
    polls adds a 0.3-0.5 ms tail to 10-15 % of the 2^22-row proofs (measured: mean 9.24 -> 9.14 ms, p90 9.50 -> 9.20 ms under taskset) — what
    any deployment does with numactl. Only a core of the GPU's own NUMA node is taken (a far core costs more than the jitter: measured); when
    the node cannot be determined nothing is pinned. Opt-in (--pin): on other boxes of the pool the same pinning changed nothing or cost 1 %."""
    cpu = None

    def __init__(self, enabled, device=0, local_rank=0, world=1):
        self.enabled, self.device, self.local_rank, self.world, self.old = enabled, device, local_rank, world, None

    def __enter__(self):
        if not self.enabled or not hasattr(os, "sched_setaffinity"):
            return self
        try:
            old = os.sched_getaffinity(0)
            near = gpu_local_cpus(self.device)
            cand = sorted(old & near) if near else []
            if not cand:
                return self
            # ranks that share a node take different cores; the first cores of a node are left to interrupt handling
            cpu = cand[(2 + 2 * self.local_rank) % len(cand)]
            os.sched_setaffinity(0, {cpu})
            self.old = old
            pinned_host_thread.cpu = cpu
        except OSError:
            self.old = None
        return self

    def __exit__(self, *exc):
        if self.old is not None:
            os.sched_setaffinity(0, self.old)
        return False


def rank_environments(n, port, base_env=None):
    """The environment of each of the n rank processes the self-launcher starts (what torch.distributed.run would have set)."""
    base = dict(os.environ if base_env is None else base_env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return [dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                 BFHIP_BENCH_SELF_LAUNCHED="1") for r in range(n)]


def launch_ranks(cmd, n, timeout, out=None, base_env=None, poll=0.1):
    """`python3 bench.py --gpus N` without a launcher: starts the N ranks as fresh child processes of THIS process — which has not touched the
    GPU and never will (a process that initialised the GPU must not be replaced or forked from) —, relays rank 0's one JSON line to `out`,
    ends the stragglers (the exact PIDs started here) when a rank fails or the limit passes, and returns the exit code: 0 = every rank exited
    0 and rank 0 printed its line; 1 = a rank failed or the line is missing; 124 = the limit passed."""
    import socket
    import subprocess
    import tempfile
    out = out or sys.stdout
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs, line_file = [], tempfile.NamedTemporaryFile(prefix="bfhip_bench_rank0_", suffix=".out", delete=False)
    try:
        for r, env in enumerate(rank_environments(n, port, base_env)):
            # rank 0's stdout is the contract's line; whatever another rank prints goes to stderr
            procs.append(subprocess.Popen(list(cmd), env=env, stdout=line_file if r == 0 else sys.stderr, stderr=None))
        t_end, rc = time.time() + timeout, None
        while rc is None:
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes):
                bad = next(r for r, c in enumerate(codes) if c not in (None, 0))
                print(f"bench.py: rank {bad} exited with code {codes[bad]}: ending the other ranks", file=sys.stderr)
                rc = 1
            elif all(c == 0 for c in codes):
                rc = 0
            elif time.time() > t_end:
                print(f"bench.py: the ranks did not finish within {timeout} s: ending them", file=sys.stderr)
                rc = 124
            else:
                time.sleep(poll)
        for p in procs:                      # stragglers: exactly the PIDs started above
            if p.poll() is None:
                p.terminate()
        t_kill = time.time() + 5
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        line_file.flush()
        lines = [l for l in open(line_file.name).read().splitlines() if l.strip()]
        line = next((l for l in reversed(lines) if l.lstrip().startswith("{")), None)
        if line is not None:
            print(line, file=out, flush=True)
        elif rc == 0:
            print("bench.py: rank 0 exited 0 without printing its JSON line", file=sys.stderr)
            rc = 1
        return rc
    finally:
        line_file.close()
        try:
            os.remove(line_file.name)
        except OSError:
            pass
