"""-m gpu: proofs in flight behind ONE caller thread (include/bfhip.h bfhip_pool_* / csrc/pool.hip). Every proof of a batch must be the bytes the CPU
oracle produces for that program — under every convention set, for every way the pool treats the preprocessed tree (recommitted by every proof as the
reference does, mod.rs:495-500; once per batch; kept across batches), whatever the number of workers — and a failing proof must not take the batch down."""
import hashlib
import os

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _prog(name):
    return open(os.path.join(HERE, "golden", "programs", name)).read()


# a mixed batch: sizes from 2^4 to 2^16 rows per component, with and without input, an output-only program, a program whose input is never read
MIXED = [
    ("+++>,<[>+.<-]", b"\x01"),
    ("++[-]+.", b""),
    ("++++++++++[>+++++++>++++++++++>+++>+<<<<-]>++.>+.+++++++..+++.>++.<<+++++++++++++++.>.+++.------.--------.>+.>.", b""),
    (_prog("a-bc.bf"), b"a"),
    (_prog("hello1.bf"), b""),
    ("[][]+[-]", b""),
    ("++[-]+.", b"\x01"),
    (_prog("loop.bf"), b""),
]
LMR = 18


@pytest.fixture(scope="module")
def wanted():
    return {}


def _oracle_proofs(oracle, conv, wanted, programs, lmr):
    key = (conv, lmr)
    if key not in wanted:
        wanted[key] = {}
    for code, inp in programs:
        if (code, inp) not in wanted[key]:
            wanted[key][(code, inp)] = oracle.prove(code, inp, log_max_rows=lmr)[0]
    return [wanted[key][p] for p in programs]


@pytest.mark.parametrize("k,mode", [(1, 1), (2, 0), (3, 1), (3, 2), (4, 0)])
def test_every_proof_of_a_mixed_batch_matches_the_oracle(pkg, oracle, conv, wanted, k, mode):
    want = _oracle_proofs(oracle, conv, wanted, MIXED, LMR)
    pool = pkg.Pool(0, n_in_flight=k, max_log_domain=LMR + 2, preprocessed=mode)
    traces = []
    try:
        traces = [pkg.Trace(pool.ctx(i % k), code, inp) for i, (code, inp) in enumerate(MIXED)]
        for rnd in range(3):            # batches of one pool: the kept / rebuilt preprocessed tree, the workers' reuse of their sub-contexts
            order = list(range(len(MIXED))) if rnd != 1 else list(reversed(range(len(MIXED))))
            proofs, info = pool.prove_batch([traces[i] for i in order], log_max_rows=LMR)
            assert info["statuses"] == [0] * len(order)
            for slot, i in enumerate(order):
                assert proofs[slot] == want[i], f"batch {rnd}, proof {slot} (program {i}) differs from the oracle (k={k}, preprocessed mode {mode})"
            assert info["batch_seconds"] > 0 and all(s > 0 for s in info["seconds"])
            # the setting applied: every worker that proved something took (or did not take) the batch's shared tree
            used = [pool.ctx(i).last_proof_flags() for i in range(k)]
            assert any(u["shared_preprocessed"] for u in used) == (mode != 0) and not any(u["kept_preprocessed"] for u in used), (mode, used)
        assert pkg.verify_brainfuck(proofs[0], LMR) == (True, "")
        # an empty batch is a no-op
        assert pool.prove_batch([], log_max_rows=LMR)[0] == []
    finally:
        for t in traces:
            t.close()
        pool.close()


@pytest.mark.with_poseidon
def test_batch_of_programs_matches_the_oracle(pkg, oracle, conv, wanted):
    """bfhip_prove_batch_brainfuck: VM, table build and upload inside the workers (what bfhip_prove_brainfuck does for one program)."""
    programs = MIXED[:5]
    want = _oracle_proofs(oracle, conv, wanted, programs, 16)
    pool = pkg.Pool(0, n_in_flight=2, max_log_domain=18)
    try:
        for _ in range(2):
            proofs, info = pool.prove_batch_brainfuck(programs, log_max_rows=16)
            assert proofs == want
    finally:
        pool.close()


@pytest.mark.single_conv
def test_a_failing_proof_does_not_take_the_batch_down(pkg, oracle, conv, wanted):
    """One trace of the batch exceeds LOG_MAX_ROWS (collatz needs 2^21): that proof fails with the library's message, the others are delivered with
    the oracle's bytes, the call returns -1 / raises, and the pool proves the next batch as if nothing had happened."""
    small = MIXED[:3]
    want = _oracle_proofs(oracle, conv, wanted, small, LMR)
    pool = pkg.Pool(0, n_in_flight=2, max_log_domain=LMR + 2)
    traces = []
    try:
        traces = [pkg.Trace(pool.ctx(0), c, i) for c, i in small]
        big = pkg.Trace(pool.ctx(1), _prog("collatz.bf"), b"7\n")
        traces.append(big)
        with pytest.raises(pkg.BfhipError, match=r"proof 1 of the batch: .*LOG_MAX_ROWS") as e:
            pool.prove_batch([traces[0], big, traces[1], traces[2]], log_max_rows=LMR)
        assert e.value.info["statuses"] == [0, -1, 0, 0]
        assert [e.value.proofs[0], e.value.proofs[2], e.value.proofs[3]] == want and e.value.proofs[1] is None
        proofs, _ = pool.prove_batch(traces[:3], log_max_rows=LMR)
        assert proofs == want
        # a program that reads input it was not given (machine.rs:163-169 fails there too) in a batch of programs: same contract
        with pytest.raises(pkg.BfhipError, match="proof 1 of the batch: input exhausted") as e:
            pool.prove_batch_brainfuck([small[0], (",", b""), small[1]], log_max_rows=LMR)
        assert e.value.info["statuses"] == [0, -1, 0] and [e.value.proofs[0], e.value.proofs[2]] == want[:2]
        assert pool.prove_batch_brainfuck(small, log_max_rows=LMR)[0] == want
        # a batch the pool cannot even start (LOG_MAX_ROWS beyond the pool's twiddle tree: the builder of the shared preprocessed tree refuses): an error, every
        # status -1, nothing delivered — and the pool is as usable as before
        with pytest.raises(pkg.BfhipError, match="twiddle tree too small") as e:
            pool.prove_batch(traces[:3], log_max_rows=LMR + 4)
        assert e.value.info["statuses"] == [-1, -1, -1] and e.value.proofs == [None, None, None]
        assert pool.prove_batch(traces[:3], log_max_rows=LMR)[0] == want
    finally:
        for t in traces:
            t.close()
        pool.close()


@pytest.mark.single_conv
def test_pool_proofs_equal_single_context_proofs_at_2p20_rows(pkg, conv):
    """A size at which the workers really overlap (2^20 domain rows, 168 launches per proof): 6 proofs of two different traces through a pool of 3, every
    preprocessed mode, against the same traces proved one at a time on a plain context."""
    lmr = 20
    # bench.py's synthetic nested-counter family: b = 250 puts the Memory component on exactly 2^20 domain rows
    progs = ["+" * 14 + "[>" + "+" * b + "[>+<-]<-]" for b in (250, 125)]
    c = pkg.Context(0, max_log_domain=lmr + 2)
    try:
        single = []
        for code in progs:
            t = pkg.Trace(c, code, b"")
            single.append(hashlib.sha256(t.prove(log_max_rows=lmr)[0]).hexdigest())
            t.close()
    finally:
        c.close()
    for mode in (0, 1, 2):
        pool = pkg.Pool(0, n_in_flight=3, max_log_domain=lmr + 2, preprocessed=mode)
        traces = []
        try:
            traces = [pkg.Trace(pool.ctx(0), code, b"") for code in progs]
            batch = [traces[0], traces[1], traces[0], traces[0], traces[1], traces[1]]
            for _ in range(2):
                proofs, info = pool.prove_batch(batch, log_max_rows=lmr)
                got = [hashlib.sha256(p).hexdigest() for p in proofs]
                assert got == [single[0], single[1], single[0], single[0], single[1], single[1]], (mode, got)
        finally:
            for t in traces:
                t.close()
            pool.close()


@pytest.mark.single_conv
def test_pool_under_the_blocking_sync_policy_and_with_settings_per_sub_context(pkg, oracle, conv, wanted):
    """What a deployment with more waiting threads than cores sets: bfhip_ctx_set_sync_policy(sub-context, blocking) — the workers sleep in their waits (and
    the automatic mailbox order switches itself off). Same bytes. A sub-context whose conventions were changed individually does not match the batch's shared
    preprocessed tree and commits its own (its proofs then follow ITS conventions)."""
    want = _oracle_proofs(oracle, conv, wanted, MIXED[:4], LMR)
    pool = pkg.Pool(0, n_in_flight=3, max_log_domain=LMR + 2)
    traces = []
    try:
        for i in range(3):
            pool.ctx(i).set_sync_policy(True)
        traces = [pkg.Trace(pool.ctx(0), c, i) for c, i in MIXED[:4]]
        for _ in range(2):
            proofs, _ = pool.prove_batch(traces * 2, log_max_rows=LMR)
            assert proofs == want * 2
        assert not any(pool.ctx(i).last_proof_flags()["mailbox_order"] for i in range(3))
        # one worker on its own: the RFC 7693 node hash on sub-context 2 only -> its proofs are that convention's, the others' the pool's
        pool.ctx(2).set_conventions(1, 0, 0, 0)
        proofs, _ = pool.prove_batch(traces * 3, log_max_rows=LMR)
        ok_default = [p == want[i % 4] for i, p in enumerate(proofs)]
        rfc = [pkg.verify_brainfuck(p, LMR, (1, 0, 0, 0))[0] for p in proofs]
        assert all(a != b for a, b in zip(ok_default, rfc)) and any(rfc) and any(ok_default), (ok_default, rfc)      # every proof is exactly one of the two kinds
        assert pool.ctx(2).last_proof_flags()["shared_preprocessed"] is False and pool.ctx(0).last_proof_flags()["shared_preprocessed"] is True
    finally:
        for t in traces:
            t.close()
        pool.close()


@pytest.mark.single_conv
def test_sub_contexts_share_the_twiddle_tree(pkg, conv):
    pool = pkg.Pool(0, n_in_flight=3, max_log_domain=22)
    try:
        tw = [pool.ctx(i).twiddles() for i in range(3)]
        assert tw[0] == tw[1] == tw[2]
        assert pool.ctx(0).memory()["twiddles"] == 2 * 4 << 21 and pool.ctx(1).memory()["twiddles"] == 0
        with pytest.raises(pkg.BfhipError, match="out of range"):
            pool.ctx(3)
    finally:
        pool.close()


@pytest.mark.single_conv
def test_a_failed_context_creation_releases_what_it_had_allocated(pkg, conv):
    """bf::Ctx::init throws in the middle (out of memory at the twiddle tree): streams, events, pinned buffers, staging and counters allocated before
    the throw must be released (bfhip_ctx_create deleted the object without destroy()). Free device memory is compared before and
    after 20 failed creations; each leaked 8 MiB of staging + the point tables before the fix."""
    holder = pkg.Context(0, max_log_domain=10)
    block = None
    try:
        free0, total = pkg.device_memory(0)
        need = 2 * (4 << 28)                                   # twiddle tree + inverse of max_log_domain 29
        keep_free = need // 2                                  # leave less than the twiddles need
        block = holder.malloc(free0 - keep_free)
        free1, _ = pkg.device_memory(0)
        assert free1 < need
        for _ in range(20):
            with pytest.raises(pkg.BfhipError, match="hipMalloc|out of memory|Out of memory"):
                pkg.Context(0, max_log_domain=29)
        with pytest.raises(pkg.BfhipError):
            pkg.Pool(0, n_in_flight=2, max_log_domain=29)
        free2, _ = pkg.device_memory(0)
        assert free1 - free2 < (16 << 20), f"20 failed creations leaked {(free1 - free2) >> 20} MiB of device memory"
    finally:
        if block:
            holder.free(block)
        holder.close()
    # and the device is as usable as before
    c = pkg.Context(0, max_log_domain=20)
    try:
        assert len(pkg.prove_brainfuck("+++>,<[>+.<-]", b"\x01", ctx=c, log_max_rows=18)) > 1000
    finally:
        c.close()
