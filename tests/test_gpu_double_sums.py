"""DOUBLE sums within 1 ULP of the exact sum on EVERY aggregation kernel.

north_star's tolerance for double-precision aggregates is 1 ULP. The
reference adds in input row order (SumAggregateBase.h:48-175, plain `+=`), so
no parallel order reproduces its roundings; what the library guarantees instead
is closeness to the exact sum (math.fsum): every DOUBLE sum is kept as hi + lo
against a fixed grid, in the LDS kernels, the one-atomic-per-row
HBM kernel (array and normalized-key mode, including its per-wave hot-key
combining), the radix-partitioned LDS fold (one and two levels, sliced
partitions) and the generic hash mode. The data here is NOT dyadic: money-like
values with two decimals and uniform doubles, ~100 rows per group, so that a
plain accumulation is visibly off (the last test proves the suite has that
power by switching the split off)."""
import math

import numpy as np
import pytest

from velox_amd import abi
from gpu_util import batch_of, exact_group_sums, run_agg, ulp_distance

pytestmark = pytest.mark.gpu

AGGS = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT), (abi.AGG_AVG, 1, abi.DOUBLE)]


def _money(rng, n):
    # cents as doubles: not representable, magnitudes 9e2 .. 1e5 (l_extendedprice-like)
    return rng.integers(90000, 10500000, n) / 100.0


def _check(got, keys_of_row, v, what, max_ulp=1):
    keys = [tuple(x) for x in zip(*[np.asarray(c[0]).tolist() for c in got[:len(keys_of_row)]])]
    exact = exact_group_sums(list(zip(*[np.asarray(k).tolist() for k in keys_of_row])), v)
    e = np.array([exact[k] for k in keys])
    err = ulp_distance(got[len(keys_of_row)][0], e)
    assert (err <= max_ulp).all(), "%s: max %s ULP from the exact sum" % (what, err.max())
    cnt = np.asarray(got[len(keys_of_row) + 1][0], dtype=np.float64)
    assert (ulp_distance(got[len(keys_of_row) + 2][0], e / cnt) <= max_ulp + 1).all(), what + " (avg)"
    return err


def _kernels(vx):
    return set(vx.profile().keys())


def _run(vx, batches, key_cols, key_types, **kw):
    vx.profile_reset()
    vx.profile_enable(True)
    got, op = run_agg(vx, batches, key_cols, key_types, AGGS, max_rows=1 << 20, **kw)
    vx.profile_enable(False)
    return got, op, _kernels(vx)


@pytest.mark.parametrize("values", ["money", "uniform"])
def test_hbm_atomic_kernel_array_mode(vx, values, monkeypatch):
    """k_agg_global, array mode: 20 000 groups (too many for the LDS kernels), radix path off."""
    monkeypatch.setenv("VX355_AGG_RADIX_MIN_ROWS", "-1")
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    rng = np.random.default_rng(101)
    n = 2_000_000
    k = rng.integers(0, 20000, n).astype(np.int64)
    v = _money(rng, n) if values == "money" else rng.random(n)
    got, op, kernels = _run(vx, [batch_of([k, v])], [0], [abi.BIGINT])
    assert "k_agg_global" in kernels and "k_rp_aggregate" not in kernels
    assert op.stats().hash_mode == abi.MODE_ARRAY
    _check(got, [k], v, "k_agg_global/array/" + values)


def test_hbm_atomic_kernel_normalized_key_mode_and_hot_keys(vx, monkeypatch):
    """k_agg_global, open addressing on sparse keys; a quarter of the rows sit on three keys, so
    the per-wave combining of hot keys (updateGlobalWave) adds most of their values."""
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    rng = np.random.default_rng(102)
    n = 2_000_000
    j = rng.integers(0, 20000, n).astype(np.int64)
    hot = rng.random(n) < 0.25
    j[hot] = rng.integers(0, 3, int(hot.sum()))
    k = ((j.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) % np.uint64(1 << 58)).astype(np.int64)
    v = _money(rng, n)
    got, op, kernels = _run(vx, [batch_of([k, v])], [0], [abi.BIGINT])
    assert "k_agg_global" in kernels
    assert op.stats().hash_mode == abi.MODE_NORMALIZED_KEY
    _check(got, [k], v, "k_agg_global/normalized + hot keys")


@pytest.mark.parametrize("max_bins,slice_recs", [(None, None), ("8", None), (None, "2048")])
def test_radix_partitioned_fold(vx, max_bins, slice_recs, monkeypatch):
    """k_rp_aggregate: one level, two levels (max_bins), and partitions cut into slices that are
    flushed with HBM atomics (slice_recs)."""
    monkeypatch.setenv("VX355_AGG_RADIX_MIN_ROWS", "1")
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    if max_bins:
        monkeypatch.setenv("VX355_AGG_RADIX_BINS", max_bins)
    if slice_recs:
        monkeypatch.setenv("VX355_AGG_RADIX_SLICE", slice_recs)
    rng = np.random.default_rng(103)
    n = 2_000_000
    k = rng.integers(0, 20000, n).astype(np.int64)
    v = _money(rng, n)
    valid = rng.random(n) > 0.05
    got, op, kernels = _run(vx, [batch_of([k, v], [None, valid])], [0], [abi.BIGINT])
    assert "k_rp_aggregate" in kernels and op.stats().radix_launches >= 1
    keys = [int(x) for x in got[0][0]]
    exact = exact_group_sums([(int(x),) for x in k], v, valid)
    e = np.array([exact[(kk,)] for kk in keys])
    err = ulp_distance(got[1][0], e)
    assert (err <= 1).all(), err.max()


def test_generic_hash_mode(vx, monkeypatch):
    """k_agg_generic (the reference's kHash): a DOUBLE grouping key and a 10-byte string key."""
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    rng = np.random.default_rng(104)
    n = 1_000_000
    j = rng.integers(0, 10000, n)
    kd = j.astype(np.float64) * 0.37
    v = _money(rng, n)
    got, op, kernels = _run(vx, [batch_of([kd, v])], [0], [abi.DOUBLE])
    assert "k_agg_generic" in kernels and op.stats().hash_mode == abi.MODE_HASH
    _check(got, [kd], v, "k_agg_generic/double key")
    ks = [b"key-%06d" % x for x in j.tolist()]
    got, op, kernels = _run(vx, [batch_of([ks, v])], [0], [abi.VARCHAR])
    assert "k_agg_generic" in kernels
    exact = exact_group_sums([(x,) for x in ks], v)
    e = np.array([exact[(bytes(x),)] for x in got[0][0]])
    assert (ulp_distance(got[1][0], e) <= 1).all()


def test_partial_to_final_chain_stays_within_one_ulp(vx, monkeypatch):
    """Four PARTIAL operators over row shards (what N Drivers or N GPUs do), one FINAL merge
    of their intermediate rows: still within 1 ULP of the exact sum of all rows."""
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    rng = np.random.default_rng(105)
    n = 2_000_000
    k = rng.integers(0, 20000, n).astype(np.int64)
    v = _money(rng, n)
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    parts = []
    for s in range(4):
        lo, hi = s * n // 4, (s + 1) * n // 4
        out, _ = run_agg(vx, [batch_of([k[lo:hi], v[lo:hi]])], [0], [abi.BIGINT], aggs, abi.STEP_PARTIAL,
                         max_rows=1 << 20)
        parts.append(out)
    pk = np.concatenate([np.asarray(p[0][0], dtype=np.int64) for p in parts])
    ps = np.concatenate([np.asarray(p[1][0], dtype=np.float64) for p in parts])
    pc = np.concatenate([np.asarray(p[2][0], dtype=np.int64) for p in parts])
    fin_aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, 2, abi.BIGINT)]
    got, _ = run_agg(vx, [batch_of([pk, ps, pc])], [0], [abi.BIGINT], fin_aggs, abi.STEP_FINAL, max_rows=1 << 20)
    exact = exact_group_sums([(int(x),) for x in k], v)
    e = np.array([exact[(int(kk),)] for kk in got[0][0]])
    # each partial is within 1 ULP of ITS exact sum, the merge adds them on a grid again:
    # the total stays within 1 ULP of the exact total for same-sign data.
    assert (ulp_distance(got[1][0], e) <= 1).all()
    cnt = np.bincount(k, minlength=20000)
    assert (np.asarray(got[2][0]) == cnt[np.asarray(got[0][0])]).all()


def test_mixed_sign_values_are_bounded_by_the_magnitude_sum(vx, monkeypatch):
    """With cancellation "1 ULP of the result" is not attainable by any fixed-width accumulator;
    the guarantee is |error| <= 1 ULP of sum(|v|) (hi is exact, lo's error is ~2^-21 of that)."""
    monkeypatch.setenv("VX355_AGG_RADIX_MIN_ROWS", "-1")
    rng = np.random.default_rng(106)
    n = 1_000_000
    k = rng.integers(0, 10000, n).astype(np.int64)
    v = _money(rng, n) * rng.choice([-1.0, 1.0], n)
    got, _, _ = _run(vx, [batch_of([k, v])], [0], [abi.BIGINT])
    exact = exact_group_sums([(int(x),) for x in k], v)
    mag = exact_group_sums([(int(x),) for x in k], np.abs(v))
    for kk, s in zip(got[0][0], got[1][0]):
        assert abs(s - exact[(int(kk),)]) <= np.spacing(mag[(int(kk),)])


def test_the_suite_detects_plain_accumulation(vx, monkeypatch):
    """Power check: with the split switched off (VX355_EXACT_SUMS=0) the same data is more than
    1 ULP away from the exact sum in the HBM-atomic kernel — i.e. the tests above would fail if a
    kernel accumulated plainly."""
    monkeypatch.setenv("VX355_AGG_RADIX_MIN_ROWS", "-1")
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    monkeypatch.setenv("VX355_EXACT_SUMS", "0")
    rng = np.random.default_rng(101)
    n = 2_000_000
    k = rng.integers(0, 20000, n).astype(np.int64)
    v = _money(rng, n)
    got, _, kernels = _run(vx, [batch_of([k, v])], [0], [abi.BIGINT])
    assert "k_agg_global" in kernels
    exact = exact_group_sums([(int(x),) for x in k], v)
    e = np.array([exact[(int(kk),)] for kk in got[0][0]])
    assert ulp_distance(got[1][0], e).max() > 1
