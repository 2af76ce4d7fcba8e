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


@pytest.mark.parametrize("log", [3, 4, 5, 6, 7, 9, 11, 12, 13, 14, 16, 19, 20])
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
