"""-m gpu: one proof over several contexts ("shard group", include/bfhip.h: bfhip_ctx_set_shard). The group members run on the one
GPU of the test box — one context, stream and host thread each — and exchange through an in-process all-gather / max-reduce, which is
what the callbacks see from torch.distributed on a multi-GPU node. Every rank must produce exactly the single-GPU proof."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PROGS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "programs")


class InProcessGroup:
    """Rendezvous of `count` threads: allgather / allreduce_max with torch.distributed semantics."""

    def __init__(self, count):
        self.count = count
        self.barrier = threading.Barrier(count)
        self.slots = [None] * count
        self.calls = {"allgather": 0, "allreduce": 0}

    def exchanges(self, rank):
        def allgather(send: bytes) -> bytes:
            self.slots[rank] = send
            self.barrier.wait()
            out = b"".join(self.slots)
            self.barrier.wait()
            if rank == 0:
                self.calls["allgather"] += 1
            return out

        def allreduce_max(values):
            self.slots[rank] = values
            self.barrier.wait()
            out = np.maximum.reduce(self.slots)
            self.barrier.wait()
            if rank == 0:
                self.calls["allreduce"] += 1
            return out

        return allgather, allreduce_max


def _prove_sharded(pkg, code, inp, lmr, count):
    group = InProcessGroup(count)
    ctxs = [pkg.Context(0, max_log_domain=lmr + 2) for _ in range(count)]
    proofs, errors = [None] * count, []

    def run(rank):
        try:
            ctxs[rank].set_shard(rank, count, *group.exchanges(rank))
            proofs[rank] = pkg.prove_brainfuck(code, inp, ctx=ctxs[rank], log_max_rows=lmr)
        except Exception as e:      # a failing rank must not leave the others waiting at the rendezvous
            errors.append(e)
            group.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(count)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for c in ctxs:
        c.close()
    assert not errors, errors
    return proofs, group.calls


@pytest.mark.parametrize("count", [2, 4, 8])
@pytest.mark.parametrize("name,inp,lmr", [("hello_kakarot.bf", b"", 17), ("collatz.bf", b"7\n", 21)])
def test_shard_group_proof_equals_single_gpu_proof(pkg, ctx, oracle, name, inp, lmr, count):
    code = open(os.path.join(PROGS, name)).read()
    single = pkg.prove_brainfuck(code, inp, ctx=pkg.Context(0, max_log_domain=lmr + 2), log_max_rows=lmr)
    proofs, calls = _prove_sharded(pkg, code, inp, lmr, count)
    assert all(p == single for p in proofs)
    assert calls["allgather"] > 0 and calls["allreduce"] > 0          # the trees really were hashed share-wise
    assert oracle.verify(single, lmr)[0]


def test_fib19_full_size_in_a_shard_group_of_two(pkg):
    """The benchmark workload: both ranks reproduce the committed digest of the oracle's proof."""
    import hashlib, json
    code = open(os.path.join(PROGS, "fib19.bf")).read()
    proofs, _ = _prove_sharded(pkg, code, b"", 24, 2)
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fib19_lmr24_oracle_proof.json")))["stwo"]
    for p in proofs:
        assert hashlib.sha256(p).hexdigest() == want["sha256"]


def test_shard_arguments_are_checked(pkg, ctx):
    with pytest.raises(pkg.BfhipError, match="power of two"):
        ctx.set_shard(0, 3, lambda b: b, lambda v: v)
    with pytest.raises(pkg.BfhipError, match="rank"):
        ctx.set_shard(2, 2, lambda b: b, lambda v: v)
    ctx.set_shard(0, 1)


def test_independent_contexts_prove_concurrently(pkg, oracle):
    """Two contexts on one GPU driven from two host threads at the same time (bench.py --inflight): each proves its own program and
    gets exactly the proof it gets alone — no shared mutable state between contexts."""
    jobs = [("hello_kakarot.bf", b"", 17), ("collatz.bf", b"7\n", 21)]
    codes = [open(os.path.join(PROGS, n)).read() for n, _, _ in jobs]
    alone = []
    for (name, inp, lmr), code in zip(jobs, codes):
        c = pkg.Context(0, max_log_domain=lmr + 2)
        alone.append(pkg.prove_brainfuck(code, inp, ctx=c, log_max_rows=lmr))
        c.close()
    ctxs = [pkg.Context(0, max_log_domain=lmr + 2) for _, _, lmr in jobs]
    out, errors = [[None] * 4, [None] * 4], []

    def run(k):
        try:
            for r in range(4):
                out[k][r] = pkg.prove_brainfuck(codes[k], jobs[k][1], ctx=ctxs[k], log_max_rows=jobs[k][2])
        except Exception as e:
            errors.append(e)

    threads = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for c in ctxs:
        c.close()
    assert not errors, errors
    for k in range(2):
        assert all(p == alone[k] for p in out[k])
