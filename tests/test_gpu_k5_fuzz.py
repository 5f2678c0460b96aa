"""Randomised differential test of K5 (per-position quality histogram) against the oracle: read length, read count, where the
column's first byte sits, number of chunks and how they differ are all drawn at random, so every device-side path decision
(A for any uniform length and alignment, B, G) and every edge (columns inside one 256-byte row, edge rows, table copies for
short periods, the 5-plane table, trailing bytes) is hit with inputs nobody chose by hand."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _column(rng, n, L, shift, ragged, lmax):
    lens = rng.integers(0, lmax + 1, n) if ragged else np.full(n, L)
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum(lens)
    total = int(off[-1])
    data = rng.integers(33, 75, total + shift, dtype=np.uint8)
    if total:
        k = int(rng.integers(0, 1 + total // 2000))
        if k:
            data[shift + rng.integers(0, total, k)] = rng.integers(0, 256, k)
    data[:shift] = 222
    return (off + shift).astype(np.int32), data


@pytest.mark.parametrize("seed", range(12))
def test_k5_random_columns_match_the_oracle(ctx, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    for case in range(10):
        L = int(rng.choice([rng.integers(1, 330), rng.integers(30, 160), rng.choice([36, 50, 51, 75, 76, 100, 101, 125, 150, 151, 250, 251, 301])]))
        lmax = int(L + rng.choice([0, 0, 1, 17]))
        nchunks = int(rng.choice([1, 1, 2, 3, 5]))
        big = rng.random() < 0.25
        kind = rng.choice(["uniform", "uniform", "uniform", "one_ragged", "mixed_len"])
        specs = []
        for c in range(nchunks):
            n = int(rng.integers(0, 1_500_000 // max(L, 8) * (8 if big else 1) + 2))
            shift = int(rng.choice([0, 0, 4, 16, 64, rng.integers(0, 300)]))
            ragged = kind == "one_ragged" and c == nchunks - 1
            Lc = int(rng.integers(1, lmax + 1)) if (kind == "mixed_len" and c > 0) else L
            specs.append((n, Lc, shift, ragged))
        want = np.zeros((lmax, 256), np.int64)
        chunks = []
        for n, Lc, shift, ragged in specs:
            off, data = _column(rng, n, Lc, shift, ragged, lmax)
            chunks.append((ctx.to_device(off), ctx.to_device(np.concatenate([data, np.full(320, 223, np.uint8)])), n))
            if n:
                want += oracle.c5_qual_pos_hist(off, data, lmax)[0]
        d = ctx.zeros(np.int64, lmax * 256)
        ctx.qual_pos_hist_chunks(chunks, lmax, d)
        ctx.sync()
        got = d.to_host().reshape(lmax, 256)
        assert np.array_equal(got, want), (seed, case, L, lmax, specs)
