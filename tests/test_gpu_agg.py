"""HashAggregation on the MI355X vs the CPU oracle, through the C ABI.

Parity bar (BASELINE.json north_star): group keys, group order (first seen),
counts, integer sums, min/max bit-exact. DOUBLE sums are order dependent: on
exactly representable data (dyadic rationals) they must be bit-identical; on
arbitrary doubles the GPU result must lie within 1 ULP of the correctly
rounded sum (math.fsum), which is the tolerance north_star states."""
import math
import os

import numpy as np
import pytest

from velox_amd import abi
from gpu_util import assert_columns_equal, batch_of, exact_group_sums, run_agg, ulp_distance

pytestmark = pytest.mark.gpu

C1_AGGS = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT),
           (abi.AGG_MIN, 1, abi.DOUBLE), (abi.AGG_MAX, 1, abi.DOUBLE), (abi.AGG_AVG, 1, abi.DOUBLE)]


def _dyadic(rng, n, scale=1024, hi=1 << 20):
    return rng.integers(0, hi, n).astype(np.float64) / scale


def test_c1_shape_batches_bit_exact(oracle, vx):
    """BASELINE config 1 shape: k BIGINT in [0,1000), v DOUBLE, 10 000-row batches."""
    rng = np.random.default_rng(1)
    n = 200000
    k = rng.integers(0, 1000, n).astype(np.int64)
    v = _dyadic(rng, n)
    batches = [batch_of([k[i:i + 10000], v[i:i + 10000]]) for i in range(0, n, 10000)]
    exp, eop = run_agg(oracle, batches, [0], [abi.BIGINT], C1_AGGS)
    got, gop = run_agg(vx, batches, [0], [abi.BIGINT], C1_AGGS)
    assert_columns_equal(got, exp, gop.kinds, what="c1")
    st = gop.stats()
    assert st.num_groups == 1000 and st.hash_mode == abi.MODE_ARRAY and st.input_rows == n
    # first-seen order
    _, first = np.unique(k, return_index=True)
    assert list(got[0][0]) == list(k[np.sort(first)])


def test_random_doubles_within_one_ulp_of_exact_sum(oracle, vx):
    rng = np.random.default_rng(2)
    n = 300000
    k = rng.integers(0, 500, n).astype(np.int64)
    v = rng.random(n)
    got, gop = run_agg(vx, [batch_of([k, v])], [0], [abi.BIGINT], C1_AGGS)
    exp, _ = run_agg(oracle, [batch_of([k, v])], [0], [abi.BIGINT], C1_AGGS)
    assert (got[0][0] == exp[0][0]).all()
    exact = exact_group_sums([(int(x),) for x in k], v)
    keys = [int(x) for x in got[0][0]]
    e = np.array([exact[(kk,)] for kk in keys])
    # Tolerance (north_star: 1 ULP). The GPU keeps every DOUBLE sum as hi + lo
    # where hi collects the values rounded to a fixed grid (exact, order
    # independent) and lo the exact remainders, so the result is the correctly
    # rounded sum up to the tiny error of the lo accumulation: within 1 ULP of
    # math.fsum. The reference's own sequential sum is several ULP away.
    gpu_err = ulp_distance(got[1][0], e)
    assert (gpu_err <= 1).all(), gpu_err.max()
    cnt = np.asarray(got[2][0], dtype=np.float64)
    assert (got[2][0] == exp[2][0]).all()
    assert (got[3][0] == exp[3][0]).all() and (got[4][0] == exp[4][0]).all()
    assert (ulp_distance(got[5][0], e / cnt) <= 2).all()  # avg = sum/count: one more rounding


def test_two_keys_nulls_int_sums_and_ignore_null_keys(oracle, vx):
    rng = np.random.default_rng(8)
    n = 50000
    k1 = rng.integers(-300, 300, n).astype(np.int64)
    k2 = rng.integers(-5, 5, n).astype(np.int32)
    valid1 = rng.random(n) > 0.1
    v = rng.integers(-1000, 1000, n).astype(np.int64)
    vvalid = rng.random(n) > 0.2
    b = batch_of([k1, k2, v], [valid1, None, vvalid])
    aggs = [(abi.AGG_SUM, 2, abi.BIGINT), (abi.AGG_COUNT, 2, abi.BIGINT), (abi.AGG_AVG, 2, abi.BIGINT),
            (abi.AGG_MIN, 2, abi.BIGINT), (abi.AGG_MAX, 2, abi.BIGINT), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    for ignore in (False, True):
        exp, _ = run_agg(oracle, [b], [0, 1], [abi.BIGINT, abi.INTEGER], aggs, ignore_null_keys=ignore)
        got, gop = run_agg(vx, [b], [0, 1], [abi.BIGINT, abi.INTEGER], aggs, ignore_null_keys=ignore)
        assert_columns_equal(got, exp, gop.kinds, what=f"ignore={ignore}")


@pytest.mark.parametrize("array_max", [None, "64"])
def test_sparse_keys_normalized_mode_and_growth(oracle, vx, array_max, monkeypatch):
    """Sparse two-key group-by: array mode by default on the GPU (288 GB HBM);
    VX355_ARRAY_MAX forces the open-addressing (normalized key) table."""
    if array_max:
        monkeypatch.setenv("VX355_ARRAY_MAX", array_max)
    rng = np.random.default_rng(3)
    n = 120000
    k1 = (rng.integers(0, 4000, n) * 1003).astype(np.int64)
    k2 = rng.integers(-50, 50, n).astype(np.int32)
    v = _dyadic(rng, n)
    batches = [batch_of([k1[i:i + 30000], k2[i:i + 30000], v[i:i + 30000]]) for i in range(0, n, 30000)]
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT), (abi.AGG_MAX, 2, abi.DOUBLE)]
    exp, _ = run_agg(oracle, batches, [0, 1], [abi.BIGINT, abi.INTEGER], aggs)
    got, gop = run_agg(vx, batches, [0, 1], [abi.BIGINT, abi.INTEGER], aggs)
    assert_columns_equal(got, exp, gop.kinds, what="sparse")
    st = gop.stats()
    if array_max:
        assert st.hash_mode == abi.MODE_NORMALIZED_KEY
    assert st.num_groups == len(exp[0][0])


def test_range_widening_replays_deferred_rows(oracle, vx, monkeypatch):
    """Keys drift upward batch after batch (like l_orderkey): ranges widen, the
    table is re-keyed on device and only the out-of-range rows are replayed."""
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")  # one launch per batch
    rng = np.random.default_rng(4)
    batches = []
    for i in range(8):
        lo = i * 700
        k = rng.integers(lo, lo + 1000, 20000).astype(np.int64)
        s = [bytes([65 + int(x)]) for x in rng.integers(0, 3 + i, 20000)]
        batches.append(batch_of([k, s, _dyadic(rng, 20000)]))
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, batches, [0, 1], [abi.BIGINT, abi.VARCHAR], aggs)
    got, gop = run_agg(vx, batches, [0, 1], [abi.BIGINT, abi.VARCHAR], aggs)
    assert_columns_equal(got, exp, gop.kinds, what="widening")
    st = gop.stats()
    assert st.deferred_rows > 0 and st.num_rehashes > 0


@pytest.mark.parametrize("device_resident", [False, True])
def test_deferred_list_overflow_rescans_the_chunk(oracle, vx, device_resident, monkeypatch):
    """More out-of-range rows than the deferred list holds: the chunk is
    rescanned for exactly the rows the first launch could not place."""
    monkeypatch.setenv("VX355_AGG_DEFER_CAP", "100")
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    rng = np.random.default_rng(44)
    batches = []
    for i in range(4):
        n = 30000
        k = rng.integers(i * 5000, i * 5000 + 3000, n).astype(np.int64)
        k2 = rng.integers(0, 3, n).astype(np.int32)
        batches.append(batch_of([k, k2, _dyadic(rng, n)]))
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT), (abi.AGG_MIN, 2, abi.DOUBLE)]
    exp, _ = run_agg(oracle, batches, [0, 1], [abi.BIGINT, abi.INTEGER], aggs)
    inp = [vx.to_device(b) for b in batches] if device_resident else batches
    got, gop = run_agg(vx, inp, [0, 1], [abi.BIGINT, abi.INTEGER], aggs)
    assert_columns_equal(got, exp, gop.kinds, what="rescan")
    assert gop.stats().deferred_rows > 100


def test_q1_shape_string_keys_dictionary_inputs(oracle, vx):
    """TPC-H Q1 at the operator boundary: two 1-char VARCHAR keys and DOUBLE
    inputs wrapped in dictionaries over the filter's selected rows."""
    rng = np.random.default_rng(5)
    n = 100000
    rf = [bytes([c]) for c in rng.choice(list(b"ANR"), n)]
    ls = [bytes([c]) for c in rng.choice(list(b"FO"), n)]
    qty = rng.integers(1, 51, n).astype(np.float64)
    price = rng.integers(90000, 10500000, n).astype(np.float64) / 128
    disc = rng.integers(0, 11, n).astype(np.float64) / 64
    sel = np.flatnonzero(rng.random(n) < 0.985).astype(np.int32)
    m = len(sel)
    disc_price = (price * (1 - disc))[sel]

    def wrap(kind, base):
        return abi.HostColumn(kind, base, encoding=abi.DICTIONARY, indices=sel)
    b = abi.HostBatch([wrap(abi.VARCHAR, rf), wrap(abi.VARCHAR, ls), wrap(abi.DOUBLE, qty),
                       wrap(abi.DOUBLE, price), wrap(abi.DOUBLE, disc),
                       abi.HostColumn(abi.DOUBLE, disc_price)], m)
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_SUM, 3, abi.DOUBLE), (abi.AGG_SUM, 5, abi.DOUBLE),
            (abi.AGG_AVG, 2, abi.DOUBLE), (abi.AGG_AVG, 3, abi.DOUBLE), (abi.AGG_AVG, 4, abi.DOUBLE),
            (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, [b], [0, 1], [abi.VARCHAR, abi.VARCHAR], aggs)
    got, gop = run_agg(vx, [b], [0, 1], [abi.VARCHAR, abi.VARCHAR], aggs)
    assert len(got[0][0]) == 6
    assert_columns_equal(got, exp, gop.kinds, what="q1")


def test_global_aggregation_masks_and_empty_input(oracle, vx):
    v = np.array([1.0, 2.0, 3.0, 4.0])
    mcol = abi.HostColumn(abi.BOOLEAN, [True, False, True, True], valid=[True, True, False, True])
    b = abi.HostBatch([abi.HostColumn(abi.DOUBLE, v), mcol])
    aggs = [(abi.AGG_SUM, 0, abi.DOUBLE, 1), (abi.AGG_COUNT_STAR, -1, abi.BIGINT, 1),
            (abi.AGG_COUNT_STAR, -1, abi.BIGINT), (abi.AGG_MIN, 0, abi.DOUBLE, 1)]
    for batches in ([b], []):
        exp, _ = run_agg(oracle, batches, [], [], aggs)
        got, gop = run_agg(vx, batches, [], [], aggs)
        assert_columns_equal(got, exp, gop.kinds, what=f"global {len(batches)}")
    assert got[0][1][0] == False and got[1][0][0] == 0  # noqa: E712  NULL sum, zero count


def test_sum_bigint_overflow_is_a_user_error(oracle, vx):
    big = np.array([2 ** 62, 2 ** 62, 5], dtype=np.int64)
    op = vx.Aggregation([], [], [(abi.AGG_SUM, 0, abi.BIGINT)])
    op.add_input(batch_of([big]))
    op.no_more_input()
    with pytest.raises(vx.Vx355Error) as e:
        op.get_output(16)   # the 128-bit total is checked when it is read out (order-independent rule)
    assert e.value.status == abi.EUSER and "integer overflow" in str(e.value)


def test_partial_then_final_equals_single(oracle, vx):
    rng = np.random.default_rng(10)
    n = 60000
    k = rng.integers(0, 97, n).astype(np.int64)
    v = _dyadic(rng, n)
    vvalid = rng.random(n) > 0.3
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_AVG, 1, abi.DOUBLE), (abi.AGG_COUNT, 1, abi.DOUBLE),
            (abi.AGG_MIN, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    single, sop = run_agg(vx, [batch_of([k, v], [None, vvalid])], [0], [abi.BIGINT], aggs)
    fin_aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_AVG, 2, abi.DOUBLE, -1, 3),
                (abi.AGG_COUNT, 4, abi.DOUBLE), (abi.AGG_MIN, 5, abi.DOUBLE),
                (abi.AGG_COUNT_STAR, 6, abi.BIGINT)]
    for impl in (oracle, vx):
        fin = impl.Aggregation([0], [abi.BIGINT], fin_aggs, abi.STEP_FINAL)
        for lo in range(0, n, 20000):
            sl = slice(lo, lo + 20000)
            p, pop = run_agg(vx, [batch_of([k[sl], v[sl]], [None, vvalid[sl]])], [0], [abi.BIGINT], aggs,
                             step=abi.STEP_PARTIAL)
            po, _ = run_agg(oracle, [batch_of([k[sl], v[sl]], [None, vvalid[sl]])], [0], [abi.BIGINT],
                            aggs, step=abi.STEP_PARTIAL)
            assert_columns_equal(p, po, pop.kinds, what="partial")
            cols = [abi.HostColumn(abi.BIGINT, p[0][0]), abi.HostColumn(abi.DOUBLE, p[1][0], p[1][1]),
                    abi.HostColumn(abi.DOUBLE, p[2][0], p[2][1]), abi.HostColumn(abi.BIGINT, p[3][0], p[3][1]),
                    abi.HostColumn(abi.BIGINT, p[4][0]), abi.HostColumn(abi.DOUBLE, p[5][0], p[5][1]),
                    abi.HostColumn(abi.BIGINT, p[6][0])]
            fin.add_input(abi.HostBatch(cols))
        fin.no_more_input()
        final = impl.collect_output(fin)
        assert_columns_equal(final, single, sop.kinds, what=f"final via {impl.__name__}")


def test_small_types_bool_keys_real_inputs_nan_minmax(oracle, vx):
    rng = np.random.default_rng(11)
    n = 40000
    kb = rng.random(n) > 0.5
    kt = rng.integers(-128, 128, n).astype(np.int8)
    ks = rng.integers(-3, 3, n).astype(np.int16)
    r = (rng.integers(0, 1 << 12, n) / 16).astype(np.float32)
    d = rng.choice([0.5, -1.25, np.nan, np.inf, -np.inf, 1e300], n)
    small = rng.integers(-100, 100, n).astype(np.int8)
    flag = rng.random(n) > 0.5
    b = batch_of([kb, kt, ks, r, d, small, flag], [rng.random(n) > 0.1, None, None, rng.random(n) > 0.1,
                                                   rng.random(n) > 0.1, None, None])
    aggs = [(abi.AGG_SUM, 3, abi.REAL), (abi.AGG_AVG, 3, abi.REAL), (abi.AGG_MIN, 3, abi.REAL),
            (abi.AGG_MIN, 4, abi.DOUBLE), (abi.AGG_MAX, 4, abi.DOUBLE), (abi.AGG_SUM, 5, abi.TINYINT),
            (abi.AGG_MAX, 5, abi.TINYINT), (abi.AGG_MIN, 6, abi.BOOLEAN), (abi.AGG_MAX, 6, abi.BOOLEAN),
            (abi.AGG_COUNT, 4, abi.DOUBLE)]
    keys, kinds = [0, 1, 2], [abi.BOOLEAN, abi.TINYINT, abi.SMALLINT]
    exp, _ = run_agg(oracle, [b], keys, kinds, aggs)
    got, gop = run_agg(vx, [b], keys, kinds, aggs)
    assert_columns_equal(got, exp, gop.kinds, what="small types")


def test_constant_and_dictionary_keys_and_chunking(oracle, vx, monkeypatch):
    monkeypatch.setenv("VX355_AGG_CHUNK_ROWS", "4096")
    rng = np.random.default_rng(12)
    n = 50000
    base = rng.integers(-40, 40, 64).astype(np.int64)
    idx = rng.integers(0, 64, n).astype(np.int32)
    kd = abi.HostColumn(abi.BIGINT, base, rng.random(n) > 0.05, encoding=abi.DICTIONARY, indices=idx)
    kc = abi.HostColumn(abi.INTEGER, np.array([7], dtype=np.int32), encoding=abi.CONSTANT)
    v = abi.HostColumn(abi.DOUBLE, _dyadic(rng, n), rng.random(n) > 0.5)
    cv = abi.HostColumn(abi.DOUBLE, np.array([0.25]), encoding=abi.CONSTANT)
    b = abi.HostBatch([kd, kc, v, cv], n)
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_SUM, 3, abi.DOUBLE), (abi.AGG_COUNT, 2, abi.DOUBLE),
            (abi.AGG_AVG, 2, abi.DOUBLE)]
    exp, _ = run_agg(oracle, [b], [0, 1], [abi.BIGINT, abi.INTEGER], aggs)
    got, gop = run_agg(vx, [b], [0, 1], [abi.BIGINT, abi.INTEGER], aggs, max_rows=10)
    assert_columns_equal(got, exp, gop.kinds, what="encodings")


def test_high_cardinality_hbm_path(oracle, vx):
    """BASELINE config 4 shape, scaled down: many distinct BIGINT keys."""
    rng = np.random.default_rng(13)
    n = 400000
    k = rng.integers(0, 150000, n).astype(np.int64)
    v = _dyadic(rng, n)
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, [batch_of([k, v])], [0], [abi.BIGINT], aggs, max_rows=100000)
    got, gop = run_agg(vx, [batch_of([k, v])], [0], [abi.BIGINT], aggs, max_rows=100000)
    assert_columns_equal(got, exp, gop.kinds, what="high cardinality")
    # sparse 64-bit keys -> open addressing on the normalized key
    ks = ((k.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) % np.uint64(1 << 58)).astype(np.int64)
    exp, _ = run_agg(oracle, [batch_of([ks, v])], [0], [abi.BIGINT], aggs, max_rows=100000)
    got, gop = run_agg(vx, [batch_of([ks, v])], [0], [abi.BIGINT], aggs, max_rows=100000)
    assert gop.stats().hash_mode == abi.MODE_NORMALIZED_KEY
    assert_columns_equal(got, exp, gop.kinds, what="sparse high cardinality")


def test_generic_hash_mode_double_string_timestamp_keys(oracle, vx):
    """Keys without value ids (exec/VectorHasher.h:338-357) run in the generic
    hash mode: VectorHasher hash + tag/id slots + stored key images. 0.0 and
    -0.0 are one group, all NaNs are one group (type/FloatingPointUtil.h:100-109)."""
    rng = np.random.default_rng(9)
    n = 60000
    flags = [[b"A", b"NO", b"RETURNED", b"RETURNED-12b", b""][i] for i in rng.integers(0, 5, n)]
    d = rng.choice([0.0, -0.0, 1.5, float("nan"), -2.25, 1e300, -np.inf], n)
    r = rng.choice([0.0, 2.5, float("nan")], n).astype(np.float32)
    ts = np.stack([rng.integers(0, 3, n), rng.integers(0, 2, n) * 1000], axis=1)
    v = _dyadic(rng, n)
    cols = [abi.HostColumn(abi.VARCHAR, flags, rng.random(n) > 0.05), abi.HostColumn(abi.DOUBLE, d, rng.random(n) > 0.05),
            abi.HostColumn(abi.REAL, r), abi.HostColumn(abi.TIMESTAMP, ts), abi.HostColumn(abi.DOUBLE, v, rng.random(n) > 0.2)]
    aggs = [(abi.AGG_SUM, 4, abi.DOUBLE), (abi.AGG_MAX, 4, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT),
            (abi.AGG_AVG, 4, abi.DOUBLE)]
    b = abi.HostBatch(cols, n)
    for keys, kinds in [([0, 1], [abi.VARCHAR, abi.DOUBLE]), ([2], [abi.REAL]), ([3, 0], [abi.TIMESTAMP, abi.VARCHAR]),
                        ([1, 2, 3, 0], [abi.DOUBLE, abi.REAL, abi.TIMESTAMP, abi.VARCHAR])]:
        for ignore in (False, True):
            exp, _ = run_agg(oracle, [b], keys, kinds, aggs, ignore_null_keys=ignore)
            got, gop = run_agg(vx, [b], keys, kinds, aggs, ignore_null_keys=ignore)
            assert_columns_equal(got, exp, gop.kinds, what=f"generic {keys} ignore={ignore}")
            assert gop.stats().hash_mode == abi.MODE_HASH


def test_generic_hash_mode_wide_int_keys_growth_and_order(oracle, vx, monkeypatch):
    """Three BIGINT keys spanning the whole int64 range do not fit a 64-bit
    normalized key: decideHashMode's kHash case. Many batches -> slot rehash and
    group-row growth; group order stays first-seen."""
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    rng = np.random.default_rng(10)
    batches = []
    for i in range(6):
        n = 40000
        k1 = rng.integers(-2 ** 62, 2 ** 62, 3000)[rng.integers(0, 3000, n)].astype(np.int64)
        k2 = rng.integers(-2 ** 62, 2 ** 62, 7)[rng.integers(0, 7, n)].astype(np.int64)
        k3 = rng.integers(-2 ** 62, 2 ** 62, 5)[rng.integers(0, 5, n)].astype(np.int64)
        batches.append(batch_of([k1, k2, k3, _dyadic(rng, n)], [rng.random(n) > 0.02, None, None, None]))
    aggs = [(abi.AGG_SUM, 3, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT), (abi.AGG_MIN, 3, abi.DOUBLE)]
    exp, _ = run_agg(oracle, batches, [0, 1, 2], [abi.BIGINT] * 3, aggs, max_rows=50000)
    got, gop = run_agg(vx, batches, [0, 1, 2], [abi.BIGINT] * 3, aggs, max_rows=50000)
    assert_columns_equal(got, exp, gop.kinds, what="wide keys")
    st = gop.stats()
    assert st.hash_mode == abi.MODE_HASH and st.num_groups == len(exp[0][0]) and st.num_rehashes > 0


@pytest.mark.parametrize("start_short", [False, True])
def test_grouping_strings_longer_than_12_bytes(oracle, vx, start_short, monkeypatch):
    """Non-inline StringViews (type/StringView.h:76-77: size > 12, pointer instead of inline bytes) as
    grouping keys: generic hash mode (the reference's kHash), the group keeps a copy of the string in
    the operator's HBM arena and compares by content. Second key, nulls, strings of every length
    around the inline / prefix / word boundaries, several batches; start_short: the stream begins
    with <= 7-byte strings (array mode) and switches in the middle. Keys, first-seen order and
    aggregates must equal the oracle's."""
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    rng = np.random.default_rng(71)
    lens = [0, 1, 7, 8, 12, 13, 15, 16, 17, 23, 24, 25, 31, 40, 64, 100, 255]
    words = []
    for i in range(400):
        ln = lens[i % len(lens)]
        body = (b"%06d-" % i + bytes(rng.integers(97, 123, 300).astype(np.uint8)))[:ln]
        words.append(body)
    words = list(dict.fromkeys(words))
    # pairs that differ only past the 4-byte prefix / only in the last byte
    words += [b"common-prefix-and-then-A", b"common-prefix-and-then-B", b"x" * 40 + b"1", b"x" * 40 + b"2"]
    batches = []
    for bi in range(4):
        n = 20000
        pool = [w for w in words if len(w) <= 7] if (start_short and bi == 0) else words
        k = [pool[i] for i in rng.integers(0, len(pool), n)]
        kvalid = rng.random(n) > 0.03
        k2 = rng.integers(0, 3, n).astype(np.int32)
        v = rng.integers(-1000, 1000, n).astype(np.int64)
        d = _dyadic(rng, n)
        batches.append(batch_of([k, k2, v, d], [kvalid, None, None, None]))
    aggs = [(abi.AGG_SUM, 2, abi.BIGINT), (abi.AGG_COUNT_STAR, -1, abi.BIGINT), (abi.AGG_SUM, 3, abi.DOUBLE),
            (abi.AGG_MIN, 2, abi.BIGINT)]
    exp, eop = run_agg(oracle, batches, [0, 1], [abi.VARCHAR, abi.INTEGER], aggs, max_rows=333)
    got, gop = run_agg(vx, batches, [0, 1], [abi.VARCHAR, abi.INTEGER], aggs, max_rows=333)
    assert gop.stats().hash_mode == abi.MODE_HASH
    assert_columns_equal(got, exp, gop.kinds, what="long string keys")
    assert any(len(x) > 12 for x, ok in zip(got[0][0], got[0][1]) if ok)


def test_device_resident_input_and_output(oracle, vx):
    rng = np.random.default_rng(14)
    n = 1 << 18
    k = rng.integers(0, 1000, n).astype(np.int64)
    v = _dyadic(rng, n)
    hb = batch_of([k, v])
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, [hb], [0], [abi.BIGINT], aggs, max_rows=2048)
    got, gop = run_agg(vx, [vx.to_device(hb)], [0], [abi.BIGINT], aggs, max_rows=2048)
    assert_columns_equal(got, exp, gop.kinds, what="device input")


def _q1_scan_batch(rng, n, long_flags=False):
    flags = [b"A", b"N", b"R"] + ([b"RETURN", b"NONE"] if long_flags else [])
    rf = [flags[i] for i in rng.integers(0, len(flags), n)]
    ls = [bytes([c]) for c in rng.choice(list(b"FO"), n)]
    qty = rng.integers(1, 51, n).astype(np.float64)
    ep = rng.integers(90000, 10500000, n).astype(np.float64) / 128
    disc = rng.integers(0, 11, n).astype(np.float64) / 64
    tax = rng.integers(0, 9, n).astype(np.float64) / 64
    ship = rng.integers(8036, 10562, n).astype(np.int32)
    return batch_of([rf, ls, qty, ep, disc, tax, ship]), (rf, ls, qty, ep, disc, tax, ship)


Q1_TERMS = [(6, abi.CMP_LE, 10471)]
Q1_PROJ = [[(3, 1.0, 0.0), (4, -1.0, 1.0)], [(3, 1.0, 0.0), (4, -1.0, 1.0), (5, 1.0, 1.0)]]


def _q1_reference(oracle, scan, cols):
    """FilterProject then HashAggregation on the oracle, the way the reference
    runs the plan: pass-through columns wrapped in the selected-row dictionary."""
    rf, ls, qty, ep, disc, tax, ship = cols
    idx, proj, _ = oracle.filter_project(scan, Q1_TERMS, Q1_PROJ)

    def wrap(kind, base):
        return abi.HostColumn(kind, base, encoding=abi.DICTIONARY, indices=idx)
    b = abi.HostBatch([wrap(abi.VARCHAR, rf), wrap(abi.VARCHAR, ls), wrap(abi.DOUBLE, qty),
                       wrap(abi.DOUBLE, ep), wrap(abi.DOUBLE, disc), abi.HostColumn(abi.DOUBLE, proj[0]),
                       abi.HostColumn(abi.DOUBLE, proj[1])], len(idx))
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_SUM, 3, abi.DOUBLE), (abi.AGG_SUM, 5, abi.DOUBLE),
            (abi.AGG_SUM, 6, abi.DOUBLE), (abi.AGG_AVG, 2, abi.DOUBLE), (abi.AGG_AVG, 3, abi.DOUBLE),
            (abi.AGG_AVG, 4, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    return run_agg(oracle, [b], [0, 1], [abi.VARCHAR, abi.VARCHAR], aggs)[0]


@pytest.mark.parametrize("device_resident,long_flags,no_fast", [(False, False, False), (True, False, False),
                                                                 (True, True, False), (True, False, True)])
def test_fused_filter_project_aggregation_q1(oracle, vx, device_resident, long_flags, no_fast, monkeypatch):
    """vx355_agg_set_fused_input: the fused operator must equal FilterProject
    followed by HashAggregation. Exactly representable inputs -> bit-exact sums.
    Device-resident flat columns take the fast LDS kernel; 5/6-byte flags make
    the range too wide for LDS: the same plan then runs on the HBM table."""
    if no_fast:
        monkeypatch.setenv("VX355_AGG_NO_FAST", "1")
    rng = np.random.default_rng(77)
    n = 300000
    scan, cols = _q1_scan_batch(rng, n, long_flags)
    exp = _q1_reference(oracle, scan, cols)
    P = vx.PROJ
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_SUM, 3, abi.DOUBLE), (abi.AGG_SUM, P(0), abi.DOUBLE),
            (abi.AGG_SUM, P(1), abi.DOUBLE), (abi.AGG_AVG, 2, abi.DOUBLE), (abi.AGG_AVG, 3, abi.DOUBLE),
            (abi.AGG_AVG, 4, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    op = vx.Aggregation([0, 1], [abi.VARCHAR, abi.VARCHAR], aggs)
    op.set_fused_input(Q1_TERMS, Q1_PROJ)
    vx.profile_reset()
    vx.profile_enable(True)
    inp = vx.to_device(scan) if device_resident else scan
    op.add_input(inp)
    op.no_more_input()
    got = vx.collect_output(op, 100)
    vx.profile_enable(False)
    names = vx.profile()
    assert_columns_equal(got, exp, op.kinds, what="fused q1")
    # 6-byte flags: open addressing in HBM behind an LDS fold with a hashed slot map (few groups);
    # the shape-specialised kernel decodes keys of up to three bytes only.
    if not no_fast and not long_flags:  # host batches are staged flat into HBM first
        assert "k_agg_fast" in names
    else:
        assert "k_agg_fast" not in names and "k_agg_lds" in names


def test_c1_device_resident_takes_fast_kernel(oracle, vx):
    rng = np.random.default_rng(78)
    n = 1 << 20
    k = rng.integers(0, 1000, n).astype(np.int64)
    v = _dyadic(rng, n)
    hb = batch_of([k, v])
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, [hb], [0], [abi.BIGINT], aggs, max_rows=2048)
    vx.profile_reset()
    vx.profile_enable(True)
    db = vx.to_device(hb)
    got, gop = run_agg(vx, [db, db], [0], [abi.BIGINT], aggs, max_rows=2048)
    vx.profile_enable(False)
    assert "k_agg_fast" in vx.profile()
    # two passes over the same batch: sums and counts double exactly
    assert (np.asarray(got[1][0]) == 2 * np.asarray(exp[1][0])).all()
    assert (np.asarray(got[2][0]) == 2 * np.asarray(exp[2][0])).all()
    assert (got[0][0] == exp[0][0]).all()


def test_decimal_like_sums_few_groups_within_one_ulp(oracle, vx):
    """TPC-H-like money values (cents, not dyadic), 4 M rows into 4 hot groups
    through the shape-specialised LDS kernel: every sum must be within 1 ULP of
    the exact sum; the sequential CPU sum of the reference's algorithm is not."""
    import math
    rng = np.random.default_rng(123)
    n = 1 << 22
    k = rng.integers(0, 4, n).astype(np.int64)
    v = rng.integers(90000, 10500000, n) / 100.0
    hb = batch_of([k, v])
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    vx.profile_reset()
    vx.profile_enable(True)
    got, gop = run_agg(vx, [vx.to_device(hb)], [0], [abi.BIGINT], aggs)
    vx.profile_enable(False)
    assert "k_agg_fast" in vx.profile()
    exp, _ = run_agg(oracle, [hb], [0], [abi.BIGINT], aggs)
    assert (got[0][0] == exp[0][0]).all() and (got[2][0] == exp[2][0]).all()
    exact = np.array([math.fsum(v[k == g]) for g in got[0][0]])
    gpu_err = ulp_distance(got[1][0], exact)
    cpu_err = ulp_distance(exp[1][0], exact)
    assert (gpu_err <= 1).all(), (gpu_err, cpu_err)


def test_thousand_small_batches_are_coalesced(oracle, vx):
    """BASELINE config 1 as the reference feeds it (SimpleAggregates.cpp:33-34):
    many 10 000-row vectors. Host batches below 256 K rows are appended to host
    column buffers and launched in a few large pieces; encodings are flattened
    on the way. Results and group order must not change."""
    rng = np.random.default_rng(321)
    batches, n_each = [], 2000
    base = rng.integers(0, 1000, 300).astype(np.int64)
    for i in range(400):
        k = abi.HostColumn(abi.BIGINT, base, rng.random(n_each) > 0.02, encoding=abi.DICTIONARY,
                           indices=rng.integers(0, 300, n_each).astype(np.int32))
        flag = abi.HostColumn(abi.BOOLEAN, rng.random(n_each) > 0.5, rng.random(n_each) > 0.1)
        v = abi.HostColumn(abi.DOUBLE, _dyadic(rng, n_each), rng.random(n_each) > 0.1)
        c = abi.HostColumn(abi.DOUBLE, np.array([0.5]), encoding=abi.CONSTANT)
        s = abi.HostColumn(abi.VARCHAR, [[b"A", b"NO", b"RETURNED"][j] for j in rng.integers(0, 3, n_each)])
        batches.append(abi.HostBatch([k, flag, v, c, s], n_each))
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT), (abi.AGG_AVG, 3, abi.DOUBLE),
            (abi.AGG_MAX, 2, abi.DOUBLE), (abi.AGG_COUNT, 2, abi.DOUBLE, 1)]
    exp, _ = run_agg(oracle, batches, [0, 4, 1], [abi.BIGINT, abi.VARCHAR, abi.BOOLEAN], aggs, max_rows=4096)
    vx.profile_reset()
    vx.profile_enable(True)
    got, gop = run_agg(vx, batches, [0, 4, 1], [abi.BIGINT, abi.VARCHAR, abi.BOOLEAN], aggs, max_rows=4096)
    vx.profile_enable(False)
    assert_columns_equal(got, exp, gop.kinds, what="coalesced")
    assert gop.stats().input_rows == 400 * n_each
    launches = sum(v[1] for k, v in vx.profile().items() if k.startswith("k_agg_"))
    assert launches <= 12  # 800 K rows in ~4 flushes, not 400 launches


@pytest.mark.parametrize("jit", ["1", "0"])
def test_plan_shape_outside_the_table_is_instantiated_with_hiprtc(oracle, vx, jit, monkeypatch):
    """A flat plan whose shape is not among the ahead-of-time instantiations
    (INTEGER + BIGINT keys, DOUBLE filter, three sums, avg, count) gets its own
    instantiation of the same kernel body through hiprtc; with VX355_JIT=0 the
    generic kernel runs instead. Same results either way."""
    monkeypatch.setenv("VX355_JIT", jit)
    rng = np.random.default_rng(202)
    n = 1 << 20
    k1 = rng.integers(0, 12, n).astype(np.int32)
    k2 = rng.integers(100, 103, n).astype(np.int64)
    a, b, c = _dyadic(rng, n), _dyadic(rng, n), _dyadic(rng, n)
    hb = batch_of([k1, k2, a, b, c])
    P = vx.PROJ
    terms = [(3, abi.CMP_LT, 900.0)]
    projs = [[(2, 1.0, 0.0), (4, 1.0, 1.0)]]
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_SUM, 3, abi.DOUBLE), (abi.AGG_SUM, P(0), abi.DOUBLE),
            (abi.AGG_AVG, 4, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    # reference: FilterProject then HashAggregation on the oracle
    idx, proj, _ = oracle.filter_project(hb, terms, projs)
    ref_b = batch_of([k1[idx], k2[idx], a[idx], b[idx], c[idx], proj[0]])
    ref_aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_SUM, 3, abi.DOUBLE), (abi.AGG_SUM, 5, abi.DOUBLE),
                (abi.AGG_AVG, 4, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, [ref_b], [0, 1], [abi.INTEGER, abi.BIGINT], ref_aggs)
    op = vx.Aggregation([0, 1], [abi.INTEGER, abi.BIGINT], aggs)
    op.set_fused_input(terms, projs)
    vx.profile_reset()
    vx.profile_enable(True)
    op.add_input(vx.to_device(hb))
    op.no_more_input()
    got = vx.collect_output(op, 100)
    vx.profile_enable(False)
    assert_columns_equal(got, exp, op.kinds, what=f"jit={jit}")
    names = vx.profile()
    if jit == "1":
        assert op.stats().reserved > 0 and "k_agg_fast" in names
    else:
        assert op.stats().reserved == 0 and "k_agg_lds" in names


def test_async_instantiation_does_not_stall_the_first_batches(oracle, vx, monkeypatch):
    """VX355_JIT=async: the first batches of a plan shape outside the table run
    on the interpreting kernel while hiprtc works on a helper thread; once the
    code object is there later batches take it. Results do not depend on which
    kernel consumed which batch."""
    import time
    # (the very first compilation of a process runs on the calling thread - it constructs the compiler's
    # statics, so that exit() can wait for later background compiles before they are destroyed: agg.hip,
    # gJitWarm - so the process is warmed with another shape before the asynchronous path is timed)
    monkeypatch.setenv("VX355_JIT", "sync")
    warm_rng = np.random.default_rng(7)
    warm = vx.Aggregation([0, 1], [abi.INTEGER, abi.BIGINT],
                          [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_MIN, 3, abi.DOUBLE), (abi.AGG_MAX, 2, abi.DOUBLE),
                           (abi.AGG_COUNT_STAR, -1, abi.BIGINT)])
    warm.add_input(vx.to_device(batch_of([warm_rng.integers(0, 5, 1 << 16).astype(np.int32),
                                          warm_rng.integers(0, 3, 1 << 16).astype(np.int64),
                                          _dyadic(warm_rng, 1 << 16), _dyadic(warm_rng, 1 << 16)])))
    warm.no_more_input()
    vx.collect_output(warm, 100)
    assert warm.stats().reserved > 0, "the warm-up shape did not go through hiprtc"
    monkeypatch.setenv("VX355_JIT", "async")
    rng = np.random.default_rng(2025)
    n = 1 << 18
    k1 = rng.integers(0, 7, n).astype(np.int64)
    k2 = rng.integers(-3, 3, n).astype(np.int32)
    a, b = _dyadic(rng, n), _dyadic(rng, n)
    hb = batch_of([k1, k2, a, b])
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_SUM, 3, abi.DOUBLE), (abi.AGG_AVG, 2, abi.DOUBLE),
            (abi.AGG_AVG, 3, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    op = vx.Aggregation([0, 1], [abi.BIGINT, abi.INTEGER], aggs)
    db = vx.to_device(hb)
    t0 = time.time()
    op.add_input(db)
    first = time.time() - t0
    batches = 1
    deadline = time.time() + 60
    while op.stats().reserved == 0 and time.time() < deadline:
        time.sleep(0.05)
        op.add_input(db)
        batches += 1
    assert op.stats().reserved > 0, "the instantiation never arrived"
    assert first < 0.5, f"first add_input took {first:.2f} s"
    op.add_input(db)
    batches += 1
    op.no_more_input()
    got = vx.collect_output(op, 100)
    exp, _ = run_agg(oracle, [hb] * batches, [0, 1], [abi.BIGINT, abi.INTEGER], aggs)
    assert_columns_equal(got, exp, op.kinds, what="async jit")


def test_dictionary_wrapped_device_inputs_take_the_specialised_kernel(oracle, vx, monkeypatch):
    """The unfused Velox pipeline: FilterProject hands HashAggregation columns
    wrapped in ONE shared index vector plus flat computed columns, all in HBM.
    That shape runs on the shape-specialised kernel (IND mask; ahead-of-time instance) instead of
    falling back to the interpreting kernel."""
    monkeypatch.setenv("VX355_JIT", "sync")   # (the default compiles in the background)
    rng = np.random.default_rng(303)
    n = 400000
    scan, cols = _q1_scan_batch(rng, n)
    rf, ls, qty, ep, disc, tax, ship = cols
    idx, proj, _ = oracle.filter_project(scan, Q1_TERMS, Q1_PROJ)
    m = len(idx)

    def wrap(kind, base):
        return abi.HostColumn(kind, base, encoding=abi.DICTIONARY, indices=idx)
    host = abi.HostBatch([wrap(abi.VARCHAR, rf), wrap(abi.VARCHAR, ls), wrap(abi.DOUBLE, qty),
                          wrap(abi.DOUBLE, ep), wrap(abi.DOUBLE, disc), abi.HostColumn(abi.DOUBLE, proj[0]),
                          abi.HostColumn(abi.DOUBLE, proj[1])], m)
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_SUM, 3, abi.DOUBLE), (abi.AGG_SUM, 5, abi.DOUBLE),
            (abi.AGG_SUM, 6, abi.DOUBLE), (abi.AGG_AVG, 2, abi.DOUBLE), (abi.AGG_AVG, 3, abi.DOUBLE),
            (abi.AGG_AVG, 4, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, [host], [0, 1], [abi.VARCHAR, abi.VARCHAR], aggs)
    # device side: base columns + ONE index vector shared by the five wrapped columns
    d_idx = vx.DeviceArray(idx)
    bases = [vx.DeviceArray(c.values) for c in host.columns[:5]]
    flats = [vx.DeviceArray(proj[0]), vx.DeviceArray(proj[1])]
    kinds = [abi.VARCHAR, abi.VARCHAR, abi.DOUBLE, abi.DOUBLE, abi.DOUBLE]
    dcols = [vx.DeviceColumn.from_ptr(k, b.ptr, m, None, abi.DICTIONARY, d_idx.ptr, n) for k, b in zip(kinds, bases)]
    dcols += [vx.DeviceColumn.from_ptr(abi.DOUBLE, f.ptr, m) for f in flats]
    dev = abi.HostBatch(dcols, m)
    vx.profile_reset()
    vx.profile_enable(True)
    got, gop = run_agg(vx, [dev], [0, 1], [abi.VARCHAR, abi.VARCHAR], aggs)
    vx.profile_enable(False)
    assert_columns_equal(got, exp, gop.kinds, what="dictionary wrapped")
    # (this shape - unfused TPC-H Q1 - is in the ahead-of-time table since round 4: no hiprtc needed)
    assert "k_agg_fast" in vx.profile() and "k_agg_lds" not in vx.profile()


@pytest.mark.parametrize("ignore_null_keys", [False, True])
def test_nullable_columns_stay_on_the_specialised_kernel(oracle, vx, monkeypatch, ignore_null_keys):
    """TPC-H Q1 with nulls in a key, in the filter column and in two aggregate inputs: the plan keeps
    its shape-specialised kernel (null bitmaps are one bit per row: FastShape::NUL) instead of
    dropping to the interpreting kernel; a null key is a group of its own (or drops the row with
    ignoreNullKeys), a null filter input fails the filter, a null input skips its accumulators
    (sum / avg / count of non-null) - SimpleNumericAggregate.h:94-160 in the same pass."""
    monkeypatch.setenv("VX355_JIT", "sync")
    rng = np.random.default_rng(505)
    n = 300_000
    scan, cols = _q1_scan_batch(rng, n)
    rf, ls, qty, ep, disc, tax, ship = cols
    valid = [rng.random(n) > p for p in (0.03, 0.0, 0.0, 0.01, 0.02, 0.0, 0.015)]
    valid[1] = valid[2] = valid[5] = None
    host = batch_of(list(cols), valid)
    P = vx.PROJ
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_SUM, 3, abi.DOUBLE), (abi.AGG_SUM, P(0), abi.DOUBLE),
            (abi.AGG_SUM, P(1), abi.DOUBLE), (abi.AGG_AVG, 2, abi.DOUBLE), (abi.AGG_AVG, 3, abi.DOUBLE),
            (abi.AGG_AVG, 4, abi.DOUBLE), (abi.AGG_COUNT, 4, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    # reference: FilterProject, then HashAggregation over its output, both on the oracle
    idx, proj, pnulls = oracle.filter_project(host, Q1_TERMS, Q1_PROJ, with_nulls=True)

    def take(c, v):
        picked = [c[i] for i in idx] if isinstance(c, list) else np.asarray(c)[idx]
        return picked, (None if v is None else np.asarray(v)[idx])
    picked = [take(c, v) for c, v in zip(cols, valid)]
    ref = batch_of([p[0] for p in picked[:5]] + [proj[0], proj[1]],
                   [p[1] for p in picked[:5]] + [pnulls[0], pnulls[1]])
    ref_aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_SUM, 3, abi.DOUBLE), (abi.AGG_SUM, 5, abi.DOUBLE),
                (abi.AGG_SUM, 6, abi.DOUBLE), (abi.AGG_AVG, 2, abi.DOUBLE), (abi.AGG_AVG, 3, abi.DOUBLE),
                (abi.AGG_AVG, 4, abi.DOUBLE), (abi.AGG_COUNT, 4, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, [ref], [0, 1], [abi.VARCHAR, abi.VARCHAR], ref_aggs, ignore_null_keys=ignore_null_keys)
    op = vx.Aggregation([0, 1], [abi.VARCHAR, abi.VARCHAR], aggs, ignore_null_keys=ignore_null_keys)
    op.set_fused_input(Q1_TERMS, Q1_PROJ)
    vx.profile_reset()
    vx.profile_enable(True)
    op.add_input(vx.to_device(host))
    op.no_more_input()
    got = vx.collect_output(op, 100)
    vx.profile_enable(False)
    assert_columns_equal(got, exp, op.kinds, what="nullable fast shape")   # dyadic data: exact sums
    names = vx.profile()
    assert "k_agg_fast" in names and "k_agg_lds" not in names and op.stats().reserved > 0


@pytest.mark.parametrize("nullable", [False, True])
@pytest.mark.parametrize("plan", ["sums_counts_masks", "integer_sums_min_max"])
def test_real_and_integer_operands_and_masks_stay_on_the_specialised_kernel(oracle, vx, monkeypatch, nullable, plan):
    """Low-cardinality plans whose operands are REAL / BIGINT / INTEGER columns (every operand reaches the
    arithmetic as a double, FastShape::LK) and whose aggregates carry FILTER masks (flat BOOLEAN columns,
    FastShape::MSK; a false or null mask skips the accumulator for the row): sum(REAL), avg(BIGINT),
    avg(INTEGER), sum(DOUBLE) FILTER m1, count(*) FILTER m2, count(REAL) - on k_agg_fast, equal to the oracle
    (dyadic values: sums are exact)."""
    monkeypatch.setenv("VX355_JIT", "sync")
    rng = np.random.default_rng(606 + nullable)
    n = 250_000
    k = rng.integers(0, 40, n).astype(np.int32)
    r = (rng.integers(-4096, 4096, n) / 16.0).astype(np.float32)
    b = rng.integers(-2 ** 31, 2 ** 31, n).astype(np.int64)
    i = rng.integers(-10 ** 6, 10 ** 6, n).astype(np.int32)
    d = rng.integers(-1 << 20, 1 << 20, n) / 1024.0
    m1 = rng.random(n) > 0.4
    m2 = rng.random(n) > 0.7
    valids = [None] * 7
    if nullable:
        valids = [None, rng.random(n) > 0.05, rng.random(n) > 0.1, None, rng.random(n) > 0.02, rng.random(n) > 0.03, None]
    host = batch_of([k, r, b, i, d, m1, m2], valids)
    aggs = [(abi.AGG_SUM, 1, abi.REAL), (abi.AGG_AVG, 2, abi.BIGINT), (abi.AGG_AVG, 3, abi.INTEGER),
            (abi.AGG_SUM, 4, abi.DOUBLE, 5), (abi.AGG_COUNT_STAR, -1, abi.BIGINT, 6), (abi.AGG_COUNT, 1, abi.REAL),
            (abi.AGG_SUM, 4, abi.DOUBLE)]
    if plan == "integer_sums_min_max":
        # FastShape::OPS: checked BIGINT sums (128-bit totals in LDS), min / max on ordered images
        # (every (column, mask) pair also owns a count of its non-null rows: five pairs here)
        aggs = [(abi.AGG_SUM, 2, abi.BIGINT), (abi.AGG_MAX, 2, abi.BIGINT), (abi.AGG_SUM, 3, abi.INTEGER, 5),
                (abi.AGG_MIN, 3, abi.INTEGER, 5), (abi.AGG_MIN, 4, abi.DOUBLE), (abi.AGG_MAX, 4, abi.DOUBLE),
                (abi.AGG_MAX, 1, abi.REAL), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, [host], [0], [abi.INTEGER], aggs)
    vx.profile_reset()
    vx.profile_enable(True)
    got, op = run_agg(vx, [vx.to_device(host)], [0], [abi.INTEGER], aggs)
    vx.profile_enable(False)
    assert_columns_equal(got, exp, op.kinds, what="REAL / integer operands and masks on the fast shape")
    names = vx.profile()
    assert "k_agg_fast" in names and "k_agg_lds" not in names and op.stats().reserved > 0


@pytest.mark.parametrize("groups", [100, 500, 3000])
def test_few_keys_over_a_wide_range(oracle, vx, groups, monkeypatch):
    """One BIGINT key with a few hundred / thousand distinct values spread over a range of 2^27: by the
    reference's rule a direct-index table (99.99 % empty rows); here the cardinality sample keeps an
    open-addressing table, the LDS kernels take it with a hashed key -> slot map while the groups fit, the
    hash-partitioned radix fold beyond. Two batches (the second meets a table with groups), nulls in the
    operand, same groups in the same first-seen order as the oracle."""
    monkeypatch.setenv("VX355_JIT", "sync")
    monkeypatch.setenv("VX355_AGG_RADIX_MIN_ROWS", "1")
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    rng = np.random.default_rng(groups)
    codes = rng.integers(0, 1 << 27, groups).astype(np.int64)
    batches = []
    for _ in range(2):
        n = 400_000
        k = codes[rng.integers(0, groups, n)]
        v = _dyadic(rng, n)
        w = rng.integers(-1 << 40, 1 << 40, n).astype(np.int64)
        batches.append(batch_of([k, v, w], [None, rng.random(n) > 0.05, None]))
    # (three accumulators: the sum, the count of its non-null inputs, count(*) - what the radix records hold)
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, batches, [0], [abi.BIGINT], aggs, max_rows=100000)
    vx.profile_reset()
    vx.profile_enable(True)
    got, op = run_agg(vx, [vx.to_device(b) for b in batches], [0], [abi.BIGINT], aggs, max_rows=100000)
    vx.profile_enable(False)
    assert_columns_equal(got, exp, op.kinds, what="few keys over a wide range (%d)" % groups)
    names = vx.profile()
    assert op.stats().hash_mode == abi.MODE_NORMALIZED_KEY and "k_card_sample" in names
    if groups <= 500:
        assert "k_agg_fast" in names and "k_agg_global" not in names and "k_rp_aggregate" not in names
    else:
        assert "k_rp_aggregate" in names and "k_agg_fast" not in names


def test_distinct_without_aggregates_and_drain_in_small_pages(oracle, vx):
    """SELECT DISTINCT k1, k2 (HashAggregation.cpp:426-488 semantics at the end
    of input): no aggregates at all; output drained 7 rows at a time."""
    rng = np.random.default_rng(404)
    n = 30000
    k1 = rng.integers(0, 40, n).astype(np.int64)
    k2 = [[b"x", b"yy", b"", b"zzz"][i] for i in rng.integers(0, 4, n)]
    b = batch_of([k1, k2], [rng.random(n) > 0.1, None])
    exp, _ = run_agg(oracle, [b], [0, 1], [abi.BIGINT, abi.VARCHAR], [], max_rows=7)
    got, gop = run_agg(vx, [b], [0, 1], [abi.BIGINT, abi.VARCHAR], [], max_rows=7)
    assert_columns_equal(got, exp, gop.kinds, what="distinct")
    assert len(got[0][0]) == len(set(zip(np.where(b.columns[0].valid, k1, -1).tolist(), k2)))


@pytest.mark.parametrize("max_bins", [None, "8"])
def test_radix_partitioned_lds_path_high_cardinality(oracle, vx, max_bins, monkeypatch):
    """BASELINE config 4 ("LDS-tiled atomic path stress"): array-mode group-bys too wide
    for one workgroup's LDS are radix-partitioned by group range (one level, or two when
    max_bins forces it) and folded partition by partition. Same results, same first-seen
    order as the oracle; the second batch widens the key range (deferred rows replay)."""
    monkeypatch.setenv("VX355_AGG_RADIX_MIN_ROWS", "1")
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    if max_bins:
        monkeypatch.setenv("VX355_AGG_RADIX_BINS", max_bins)
    rng = np.random.default_rng(77)
    n = 300000
    k1 = rng.integers(40000, 110000, n).astype(np.int64)
    k2 = rng.integers(0, 150000, n).astype(np.int64)
    v = [_dyadic(rng, n), _dyadic(rng, n)]
    w = [rng.integers(-1 << 40, 1 << 40, n).astype(np.int64) for _ in range(2)]
    vvalid = [rng.random(n) > 0.1 for _ in range(2)]
    kvalid = [rng.random(n) > 0.01 for _ in range(2)]
    batches = [batch_of([k, v[i], w[i]], [kvalid[i], vvalid[i], None]) for i, k in enumerate([k1, k2])]
    for aggs in ([(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)],
                 [(abi.AGG_SUM, 2, abi.BIGINT), (abi.AGG_MIN, 1, abi.DOUBLE), (abi.AGG_COUNT, 1, abi.DOUBLE)],
                 [(abi.AGG_MAX, 2, abi.BIGINT)]):
        exp, _ = run_agg(oracle, batches, [0], [abi.BIGINT], aggs, max_rows=100000)
        got, gop = run_agg(vx, batches, [0], [abi.BIGINT], aggs, max_rows=100000)
        assert_columns_equal(got, exp, gop.kinds, what="radix %s" % (aggs,))
        st = gop.stats()
        assert st.hash_mode == abi.MODE_ARRAY and st.radix_launches >= 2 and st.deferred_rows > 0


@pytest.mark.parametrize("shape", ["uniform", "hot_keys", "nulls_and_masks"])
def test_radix_path_over_open_addressing_tables(oracle, vx, shape, monkeypatch):
    """Sparse BIGINT keys (normalized-key mode): rows are partitioned by the HOME SLOT of their group,
    folded in an LDS window of that slot range and merged into the open-addressing table with one
    findOrInsert per group (k_rp_aggregate_hashed) - several batches (the table grows and is re-keyed
    in between), hot keys (sliced partitions flush with atomics), null keys / null inputs. Same
    groups, same first-seen order, bit-exact integers and dyadic sums as the oracle."""
    monkeypatch.setenv("VX355_AGG_RADIX_MIN_ROWS", "1")
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    if shape == "hot_keys":
        monkeypatch.setenv("VX355_AGG_RADIX_SLICE", "4096")
    rng = np.random.default_rng(909)
    n = 400_000
    batches = []
    for b in range(3):
        j = rng.integers(0, 60_000 * (b + 1), n).astype(np.uint64)
        if shape == "hot_keys":
            hot = rng.random(n) < 0.4
            j[hot] = rng.integers(0, 3, int(hot.sum())).astype(np.uint64)
        k = ((j * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(0x5DEECE66D)).astype(np.int64)   # all over int64
        v = _dyadic(rng, n)
        w = rng.integers(-1 << 40, 1 << 40, n).astype(np.int64)
        valids = [None, None, None]
        if shape == "nulls_and_masks":
            valids = [rng.random(n) > 0.01, rng.random(n) > 0.1, None]
        batches.append(batch_of([k, v, w], valids))
    for aggs in ([(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)],
                 [(abi.AGG_SUM, 2, abi.BIGINT), (abi.AGG_MIN, 1, abi.DOUBLE), (abi.AGG_COUNT, 1, abi.DOUBLE)],
                 [(abi.AGG_AVG, 1, abi.DOUBLE), (abi.AGG_MAX, 2, abi.BIGINT)]):
        exp, _ = run_agg(oracle, batches, [0], [abi.BIGINT], aggs, max_rows=100000)
        vx.profile_reset()
        vx.profile_enable(True)
        got, gop = run_agg(vx, batches, [0], [abi.BIGINT], aggs, max_rows=100000)
        vx.profile_enable(False)
        assert_columns_equal(got, exp, gop.kinds, what="hashed radix %s %s" % (shape, aggs))
        st = gop.stats()
        assert st.hash_mode == abi.MODE_NORMALIZED_KEY and st.radix_launches >= 2, (st.hash_mode, st.radix_launches)
        assert "k_rp_aggregate" in vx.profile()


@pytest.mark.parametrize("shape", ["one_batch", "more_batches", "refold", "hot_keys", "nulls_and_masks", "part_of_a_batch"])
def test_dense_folds_into_an_operator_without_groups(oracle, vx, shape, monkeypatch):
    """Open-addressing mode, no groups yet, a large batch: the radix folds APPEND their groups to a plain
    array of group rows (hashFoldFlushDense) and that array is the table until a key has to be looked up
    (then it is re-keyed into a real table): one batch (statistics show capacity == groups), more batches
    behind it, a first guess of the group count that is too small (folded twice), hot keys whose partition
    is folded in slices (rows of one key merged afterwards, k_dense_merge), null keys / inputs, and a
    batch larger than one dense launch takes."""
    monkeypatch.setenv("VX355_AGG_RADIX_MIN_ROWS", "1")
    monkeypatch.setenv("VX355_AGG_DENSE_MIN_ROWS", "1")
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    if shape == "refold":
        monkeypatch.setenv("VX355_AGG_DENSE_CAP", "1000")
    if shape == "hot_keys":
        monkeypatch.setenv("VX355_AGG_RADIX_SLICE", "4096")
    if shape == "part_of_a_batch":
        monkeypatch.setenv("VX355_AGG_DENSE_MAX_ROWS", "100032")
    rng = np.random.default_rng(4242)
    n = 400_000
    batches = []
    for b in range(1 if shape in ("one_batch", "refold") else 3):
        j = rng.integers(0, 60_000 * (b + 1), n).astype(np.uint64)
        if shape == "hot_keys":
            hot = rng.random(n) < 0.4
            j[hot] = rng.integers(0, 3, int(hot.sum())).astype(np.uint64)
        k = ((j * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(0x5DEECE66D)).astype(np.int64)
        v = _dyadic(rng, n)
        w = rng.integers(-1 << 40, 1 << 40, n).astype(np.int64)
        valids = [None, None, None]
        if shape == "nulls_and_masks":
            valids = [rng.random(n) > 0.01, rng.random(n) > 0.1, None]
        batches.append(batch_of([k, v, w], valids))
    for aggs in ([(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)],
                 [(abi.AGG_SUM, 2, abi.BIGINT), (abi.AGG_MIN, 1, abi.DOUBLE), (abi.AGG_COUNT, 1, abi.DOUBLE)],
                 [(abi.AGG_AVG, 1, abi.DOUBLE), (abi.AGG_MAX, 2, abi.BIGINT)]):
        exp, _ = run_agg(oracle, batches, [0], [abi.BIGINT], aggs, max_rows=100000)
        vx.profile_reset()
        vx.profile_enable(True)
        got, gop = run_agg(vx, batches, [0], [abi.BIGINT], aggs, max_rows=100000)
        vx.profile_enable(False)
        assert_columns_equal(got, exp, gop.kinds, what="dense folds %s %s" % (shape, aggs))
        st = gop.stats()
        prof = vx.profile()
        assert st.hash_mode == abi.MODE_NORMALIZED_KEY and "k_rp_aggregate" in prof
        if shape in ("one_batch", "refold"):
            # the array of rows is the table: the groups plus the rows of the workgroups' blocks that no
            # group took (a block is at least 256 rows, a launch has at most 4 workgroups per CU)
            assert st.num_groups <= st.capacity <= st.num_groups + 256 * 4 * 256 + 4096, (st.capacity, st.num_groups)
            # a fold = the owners' launch + the launch for the other slices of split partitions
            assert prof["k_rp_aggregate"][1] == 2 * (2 if shape == "refold" else 1)
        if shape == "hot_keys":
            assert "k_dense_merge" in prof


@pytest.mark.parametrize("shape", ["sampled_slots", "small_table_overflows", "more_batches", "unordered"])
def test_hashed_folds_size_their_lds_table_from_a_sample_of_the_keys(oracle, vx, shape, monkeypatch):
    """Two scatter levels (> 1 M rows): k_rp_distinct_sample counts the distinct keys of 64 partitions and
    the launch sizes the folds' LDS tables from that (512 entries for ~30 rows per key); a table forced
    too small for the keys of its partitions (VX355_AGG_HASH_SLOTS=512 with ~700 distinct keys per
    partition: nearly every row has a key of its own) sends the overflowing records straight to rows of their own, merged afterwards
    (k_dense_merge); a second batch goes through the folds that look their groups up in the table; the
    dense array of rows has holes (a workgroup takes rows in blocks), which neither the first-seen sort
    nor the unordered listing may show."""
    monkeypatch.setenv("VX355_AGG_RADIX_MIN_ROWS", "1")
    monkeypatch.setenv("VX355_AGG_DENSE_MIN_ROWS", "1")
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    rng = np.random.default_rng(20260923)
    n = 3_000_000
    distinct = 50_000_000 if shape == "small_table_overflows" else 100_000
    if shape == "small_table_overflows":
        monkeypatch.setenv("VX355_AGG_HASH_SLOTS", "512")
    batches = []
    for b in range(2 if shape == "more_batches" else 1):
        j = rng.integers(0, distinct * (b + 1), n).astype(np.uint64)
        k = ((j * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(0x5DEECE66D)).astype(np.int64)
        batches.append(batch_of([k, _dyadic(rng, n), rng.integers(-1 << 40, 1 << 40, n).astype(np.int64)]))
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT), (abi.AGG_MIN, 2, abi.BIGINT)]
    exp, _ = run_agg(oracle, batches, [0], [abi.BIGINT], aggs, max_rows=1 << 20)
    vx.profile_reset()
    vx.profile_enable(True)
    if shape == "unordered":
        op = vx.Aggregation([0], [abi.BIGINT], aggs, abi.STEP_SINGLE, flags=abi.AGG_UNORDERED_OUTPUT)
        for b in batches:
            op.add_input(b)
        op.no_more_input()
        got = vx.collect_output(op, 1 << 20)
        go, eo = np.argsort(got[0][0], kind="stable"), np.argsort(exp[0][0], kind="stable")
        assert len(got[0][0]) == len(exp[0][0])
        for c in range(4):
            assert (np.asarray(got[c][0])[go] == np.asarray(exp[c][0])[eo]).all()
        gop = op
    else:
        got, gop = run_agg(vx, batches, [0], [abi.BIGINT], aggs, max_rows=1 << 20)
        assert_columns_equal(got, exp, gop.kinds, what="hashed folds %s" % shape)
    vx.profile_enable(False)
    prof = vx.profile()
    st = gop.stats()
    assert st.hash_mode == abi.MODE_NORMALIZED_KEY and "k_rp_aggregate" in prof
    if shape == "small_table_overflows":
        assert "k_dense_merge" in prof and "k_rp_distinct_sample" not in prof and "k_rp_bucket_sample" not in prof
    else:
        # (round 5: and one partition per sampled level-1 bucket is counted BEFORE level 2, which then runs with
        # fewer bins when the keys repeat - here ~30 rows per key)
        assert "k_rp_distinct_sample" in prof and "k_rp_bucket_sample" in prof


@pytest.mark.parametrize("keys", ["dense", "sparse"])
def test_first_seen_order_of_many_groups_without_the_library_sort(oracle, vx, keys, monkeypatch):
    """The listed (first row, group) entries of the radix folds are put into first-seen order by the
    library's own passes (k_fs_pack, two exact radix levels, k_fs_rank: a bitmap per partition) instead of
    rocPRIM - forced here for 1.2 M groups (it starts at 32 M entries); direct-index table and the dense
    folds' array of rows with its holes. Group order must be the oracle's."""
    monkeypatch.setenv("VX355_AGG_RADIX_MIN_ROWS", "1")
    monkeypatch.setenv("VX355_AGG_DENSE_MIN_ROWS", "1")
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    monkeypatch.setenv("VX355_AGG_OWN_SORT_MIN", "1")
    rng = np.random.default_rng(77)
    n = 3_000_000
    j = rng.integers(0, 1_500_000, n).astype(np.uint64)
    k = j.astype(np.int64) if keys == "dense" else ((j * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(0x5DEECE66D)).astype(np.int64)
    batches = [batch_of([k, _dyadic(rng, n)])]
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, batches, [0], [abi.BIGINT], aggs, max_rows=1 << 20)
    vx.profile_reset()
    vx.profile_enable(True)
    got, gop = run_agg(vx, batches, [0], [abi.BIGINT], aggs, max_rows=1 << 20)
    vx.profile_enable(False)
    assert_columns_equal(got, exp, gop.kinds, what="own first-seen sort, %s keys" % keys)
    prof = vx.profile()
    assert "k_fs_rank" in prof and "k_rp_aggregate" in prof, sorted(prof)


def test_radix_path_two_keys_fused_filter(oracle, vx, monkeypatch):
    """Two grouping keys (the normalized key is the partitioning key) behind a fused filter."""
    monkeypatch.setenv("VX355_AGG_RADIX_MIN_ROWS", "1")
    rng = np.random.default_rng(78)
    n = 800000
    a = rng.integers(0, 200, n).astype(np.int64)
    b = rng.integers(-100, 100, n).astype(np.int32)
    v = _dyadic(rng, n)
    d = rng.integers(0, 100, n).astype(np.int32)
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_AVG, 2, abi.DOUBLE)]
    terms = [(3, abi.CMP_LE, 70)]
    keep = d <= 70
    exp, _ = run_agg(oracle, [batch_of([a[keep], b[keep], v[keep]])], [0, 1], [abi.BIGINT, abi.INTEGER], aggs,
                     max_rows=1 << 20)
    op = vx.Aggregation([0, 1], [abi.BIGINT, abi.INTEGER], aggs, abi.STEP_SINGLE)
    op.set_fused_input(terms, [])
    op.add_input(batch_of([a, b, v, d]))
    op.no_more_input()
    got = vx.collect_output(op, 1 << 20)
    assert_columns_equal(got, exp, op.kinds, what="radix two keys")
    assert op.stats().radix_launches == 1


@pytest.mark.parametrize("defer_cap", [None, "64"])
def test_mid_stream_switch_to_generic_mode(oracle, vx, defer_cap, monkeypatch):
    """The first batches fit a normalized key (array / open-addressing mode); a later batch
    brings values whose ranges overflow 64 bits, or a string no value id can hold (8..12 bytes).
    The reference re-decides the hash mode and rehashes (HashTable.cpp:1751-1839); here the live
    groups move to the generic structures and the stream carries on. defer_cap=64 makes the
    deferred list overflow, so the outstanding rows are found by rescanning the chunk."""
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    if defer_cap:
        monkeypatch.setenv("VX355_AGG_DEFER_CAP", defer_cap)
    rng = np.random.default_rng(91)
    n = 50000
    aggs = [(abi.AGG_SUM, 3, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT), (abi.AGG_MAX, 3, abi.DOUBLE),
            (abi.AGG_AVG, 3, abi.DOUBLE)]

    def batch(lo, hi, words):
        k1 = rng.integers(lo, hi, 500)[rng.integers(0, 500, n)].astype(np.int64)
        k2 = rng.integers(-40, 40, n).astype(np.int32)
        s = [words[i] for i in rng.integers(0, len(words), n)]
        return abi.HostBatch([abi.HostColumn(abi.BIGINT, k1, rng.random(n) > 0.03), abi.HostColumn(abi.INTEGER, k2),
                              abi.HostColumn(abi.VARCHAR, s, rng.random(n) > 0.03),
                              abi.HostColumn(abi.DOUBLE, _dyadic(rng, n), rng.random(n) > 0.1)], n)
    short = [b"", b"A", b"NO", b"RETURN", b"7 bytes"]
    # (a) integer ranges overflow in the third batch
    batches = [batch(0, 1000, short), batch(-5000, 90000, short), batch(-2 ** 62, 2 ** 62, short),
               batch(-2 ** 62, 2 ** 62, short)]
    for keys, kinds in [([0, 1], [abi.BIGINT, abi.INTEGER]), ([0, 2, 1], [abi.BIGINT, abi.VARCHAR, abi.INTEGER])]:
        exp, _ = run_agg(oracle, batches, keys, kinds, aggs, max_rows=100000)
        got, gop = run_agg(vx, batches, keys, kinds, aggs, max_rows=100000)
        assert_columns_equal(got, exp, gop.kinds, what=f"switch on range overflow {keys}")
        st = gop.stats()
        assert st.hash_mode == abi.MODE_HASH and st.num_groups == len(exp[0][0])
    # (b) a string longer than 7 bytes appears in the second batch
    batches = [batch(0, 300, short), batch(0, 300, short + [b"8 bytes!", b"twelve bytes"]), batch(0, 300, short)]
    for ignore in (False, True):
        exp, _ = run_agg(oracle, batches, [2, 0], [abi.VARCHAR, abi.BIGINT], aggs, max_rows=100000,
                         ignore_null_keys=ignore)
        got, gop = run_agg(vx, batches, [2, 0], [abi.VARCHAR, abi.BIGINT], aggs, max_rows=100000,
                           ignore_null_keys=ignore)
        assert_columns_equal(got, exp, gop.kinds, what=f"switch on long string ignore={ignore}")
        assert gop.stats().hash_mode == abi.MODE_HASH


def test_single_wide_integer_key_stays_in_normalized_mode(oracle, vx, monkeypatch):
    """One BIGINT key spanning (almost) all of int64: the reference drops to kHash above
    VectorHasher::kMaxRange; here the value itself is the 64-bit normalized key (open addressing,
    one random sector per row). The two largest int64 values use reserved ids: their appearance
    moves the table to the generic mode in mid stream."""
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    rng = np.random.default_rng(93)
    n = 120000
    pool = rng.integers(-2 ** 63, 2 ** 63 - 3, 30000, dtype=np.int64)
    pool[:3] = [-2 ** 63, 2 ** 63 - 3, 0]
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT), (abi.AGG_MIN, 1, abi.DOUBLE)]

    def batch(keys):
        return abi.HostBatch([abi.HostColumn(abi.BIGINT, keys, rng.random(len(keys)) > 0.02),
                              abi.HostColumn(abi.DOUBLE, _dyadic(rng, len(keys)), rng.random(len(keys)) > 0.1)])
    small = rng.integers(-1000, 1000, n).astype(np.int64)           # first batch: a narrow range (array mode)
    batches = [batch(small), batch(pool[rng.integers(0, 30000, n)]), batch(pool[rng.integers(0, 30000, n)])]
    exp, _ = run_agg(oracle, batches, [0], [abi.BIGINT], aggs, max_rows=100000)
    got, gop = run_agg(vx, batches, [0], [abi.BIGINT], aggs, max_rows=100000)
    assert_columns_equal(got, exp, gop.kinds, what="wide single key")
    assert gop.stats().hash_mode == abi.MODE_NORMALIZED_KEY
    # the reserved values arrive: generic mode takes over, results unchanged
    top = pool[rng.integers(0, 30000, n)].copy()
    top[::97] = 2 ** 63 - 1
    top[1::89] = 2 ** 63 - 2
    batches.append(batch(top))
    exp, _ = run_agg(oracle, batches, [0], [abi.BIGINT], aggs, max_rows=100000)
    got, gop = run_agg(vx, batches, [0], [abi.BIGINT], aggs, max_rows=100000)
    assert_columns_equal(got, exp, gop.kinds, what="wide single key + reserved values")
    assert gop.stats().hash_mode == abi.MODE_HASH


@pytest.mark.parametrize("key_dtype", [np.int64, np.int32])
def test_radix_path_flat_null_free_specialisation(oracle, vx, key_dtype, monkeypatch):
    """The loads-ahead instances of the radix kernels (flat BIGINT / INTEGER key, flat 8-byte
    operands without nulls): checked sum(BIGINT), min(DOUBLE), max(BIGINT) and count(*) only."""
    monkeypatch.setenv("VX355_AGG_RADIX_MIN_ROWS", "1")
    rng = np.random.default_rng(79)
    n = 600000
    k = rng.integers(-3000, 90000, n).astype(key_dtype)
    w = rng.integers(-1 << 40, 1 << 40, n).astype(np.int64)
    v = _dyadic(rng, n)
    kind = abi.BIGINT if key_dtype == np.int64 else abi.INTEGER
    for aggs in ([(abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_MIN, 2, abi.DOUBLE), (abi.AGG_MAX, 1, abi.BIGINT)],
                 [(abi.AGG_COUNT_STAR, -1, abi.BIGINT)],
                 [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT), (abi.AGG_MAX, 2, abi.DOUBLE)]):
        exp, _ = run_agg(oracle, [batch_of([k, w, v])], [0], [kind], aggs, max_rows=100000)
        got, gop = run_agg(vx, [batch_of([k, w, v])], [0], [kind], aggs, max_rows=100000)
        assert_columns_equal(got, exp, gop.kinds, what="radix flat %s" % (aggs,))
        assert gop.stats().radix_launches >= 1


@pytest.mark.parametrize("slice_recs", [None, "4096"])
def test_radix_path_skewed_keys_and_few_partitions(oracle, vx, slice_recs, monkeypatch):
    """Partitions much larger than the rest (90 % of the rows on one key; a key range of a few
    partitions only) are folded slice by slice by many workgroups and flushed with atomics."""
    monkeypatch.setenv("VX355_AGG_RADIX_MIN_ROWS", "1")
    monkeypatch.setenv("VX355_AGG_LDS_HASHED", "0")   # (no cardinality sample: the direct-index table and its radix path)
    if slice_recs:
        monkeypatch.setenv("VX355_AGG_RADIX_SLICE", slice_recs)
    rng = np.random.default_rng(80)
    n = 700000
    k = rng.integers(0, 100000, n).astype(np.int64)
    k[rng.random(n) < 0.9] = 77777
    w = rng.integers(-1 << 30, 1 << 30, n).astype(np.int64)
    v = _dyadic(rng, n)
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT), (abi.AGG_MIN, 1, abi.BIGINT)]
    exp, _ = run_agg(oracle, [batch_of([k, w, v])], [0], [abi.BIGINT], aggs, max_rows=200000)
    got, gop = run_agg(vx, [batch_of([k, w, v])], [0], [abi.BIGINT], aggs, max_rows=200000)
    assert_columns_equal(got, exp, gop.kinds, what="radix skew")
    assert gop.stats().radix_launches >= 1
    k2 = rng.integers(0, 9000, n).astype(np.int64)      # a handful of partitions
    exp, _ = run_agg(oracle, [batch_of([k2, w, v])], [0], [abi.BIGINT], aggs, max_rows=200000)
    got, gop = run_agg(vx, [batch_of([k2, w, v])], [0], [abi.BIGINT], aggs, max_rows=200000)
    assert_columns_equal(got, exp, gop.kinds, what="radix few partitions")


@pytest.mark.parametrize("shape", ["array", "normalized", "generic"])
def test_partial_flush_emits_and_restarts(oracle, vx, shape, monkeypatch):
    """HashAggregation.cpp:191-236,293-327: a PARTIAL operator that is "full" emits its groups and
    starts over. Flushes in the middle of the stream must not lose or duplicate anything: the FINAL
    step over all flushed pages equals a SINGLE aggregation of the whole input (oracle), and every
    page lists its groups in first-seen order of the rows since the previous flush."""
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    rng = np.random.default_rng(61)
    n, pieces = 240000, 6
    if shape == "array":
        k = rng.integers(0, 3000, n).astype(np.int64)
    elif shape == "normalized":
        k = ((rng.integers(0, 3000, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) % np.uint64(1 << 58)).astype(np.int64)
    else:
        k = rng.integers(0, 3000, n).astype(np.float64) * 0.5
    ktype = abi.DOUBLE if shape == "generic" else abi.BIGINT
    v = _dyadic(rng, n)
    w = rng.integers(-1000, 1000, n).astype(np.int64)
    vvalid = rng.random(n) > 0.1
    raw = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT), (abi.AGG_AVG, 1, abi.DOUBLE),
           (abi.AGG_MIN, 2, abi.BIGINT), (abi.AGG_COUNT, 1, abi.DOUBLE)]
    op = vx.Aggregation([0], [ktype], raw, abi.STEP_PARTIAL)
    pages = []
    step = n // pieces
    for i in range(pieces):
        lo, hi = i * step, (i + 1) * step
        op.add_input(batch_of([k[lo:hi], v[lo:hi], w[lo:hi]], [None, vvalid[lo:hi], None]))
        if i % 2 == 1 and i != pieces - 1:
            op.flush()
            page = vx.collect_output(op, 777)
            first = {}
            for r in range((i - 1) * step, hi):
                first.setdefault(k[r].item(), r)
            assert list(page[0][0]) == [kk for kk, _ in sorted(first.items(), key=lambda t: t[1])]
            pages.append(page)
            assert op.stats().num_groups == 0
    op.no_more_input()
    pages.append(vx.collect_output(op, 777))
    assert op.stats().num_flushes == 2 and op.stats().table_bytes > 0
    from velox_amd import dist as vdist
    kinds = vdist.partial_kinds([ktype], raw)
    merged = []
    for c in range(len(kinds)):
        vals = np.concatenate([np.asarray(p[c][0]) for p in pages])
        valid = np.concatenate([np.asarray(p[c][1]) for p in pages])
        merged.append(abi.HostColumn(kinds[c], vals, valid))
    fin = vx.Aggregation([0], [ktype], vdist.final_aggs_for(raw, 1), abi.STEP_FINAL)
    fin.add_input(abi.HostBatch(merged))
    fin.no_more_input()
    got = vx.collect_output(fin, 5000)
    exp, eop = run_agg(oracle, [batch_of([k, v, w], [None, vvalid, None])], [0], [ktype], raw, max_rows=5000)
    order = np.argsort(np.asarray(got[0][0]), kind="stable")
    eorder = np.argsort(np.asarray(exp[0][0]), kind="stable")
    for c in range(len(exp)):
        g, e = np.asarray(got[c][0])[order], np.asarray(exp[c][0])[eorder]
        gv, ev = np.asarray(got[c][1])[order], np.asarray(exp[c][1])[eorder]
        assert (gv == ev).all() and (g[ev] == e[ev]).all(), c
    with pytest.raises(vx.Vx355Error):
        vx.Aggregation([0], [ktype], raw, abi.STEP_SINGLE).flush()


def test_to_intermediate_is_the_partial_layout_of_single_row_groups(oracle, vx):
    """Abandoned partial aggregation (HashAggregation.cpp:185-189 -> GroupingSet::toIntermediate,
    GroupingSet.cpp:1589-1675): raw rows straight into the PARTIAL layout. Checked against the rules
    of Appendix C row by row, and by feeding the result to a FINAL step: it must equal the SINGLE
    aggregation of the same rows (oracle)."""
    rng = np.random.default_rng(62)
    n = 100000
    k = rng.integers(0, 700, n).astype(np.int64)
    x = _dyadic(rng, n)
    xvalid = rng.random(n) > 0.2
    y = rng.integers(-50, 50, n).astype(np.int32)
    yvalid = rng.random(n) > 0.2
    f = rng.random(n).astype(np.float32)
    m = rng.random(n) > 0.5
    mvalid = rng.random(n) > 0.1
    b = batch_of([k, x, y, f, m], [None, xvalid, yvalid, None, mvalid])
    raw = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_SUM, 2, abi.INTEGER), (abi.AGG_COUNT_STAR, -1, abi.BIGINT),
           (abi.AGG_COUNT, 2, abi.INTEGER, 4), (abi.AGG_AVG, 1, abi.DOUBLE, 4), (abi.AGG_MIN, 2, abi.INTEGER),
           (abi.AGG_MAX, 3, abi.REAL), (abi.AGG_SUM, 3, abi.REAL)]
    from velox_amd import dist as vdist
    kinds = vdist.partial_kinds([abi.BIGINT], raw)
    op = vx.Aggregation([0], [abi.BIGINT], raw, abi.STEP_PARTIAL)
    cols = op.to_intermediate(b, kinds[1:])
    mask_ok = m & mvalid
    exp = [(x, xvalid), (y.astype(np.int64), yvalid), (np.ones(n, dtype=np.int64), np.ones(n, bool)),
           ((yvalid & mask_ok).astype(np.int64), np.ones(n, bool)), (x, xvalid & mask_ok),
           ((xvalid & mask_ok).astype(np.int64), xvalid & mask_ok), (y, yvalid), (f, np.ones(n, bool)),
           (f.astype(np.float64), np.ones(n, bool))]
    assert len(cols) == len(exp)
    for c, ((gv, gvalid), (ev, evalid)) in enumerate(zip(cols, exp)):
        assert (np.asarray(gvalid) == evalid).all(), c
        assert (np.asarray(gv)[evalid] == np.asarray(ev)[evalid]).all(), c
    fin = vx.Aggregation([0], [abi.BIGINT], vdist.final_aggs_for(raw, 1), abi.STEP_FINAL)
    fin.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, k)] +
                                [abi.HostColumn(kind, np.asarray(v), np.asarray(valid))
                                 for kind, (v, valid) in zip(kinds[1:], cols)]))
    fin.no_more_input()
    got = vx.collect_output(fin, 5000)
    exp_single, eop = run_agg(oracle, [b], [0], [abi.BIGINT], raw, max_rows=5000)
    assert_columns_equal(got, exp_single, eop.kinds, what="final(toIntermediate) vs single")


def test_unordered_output_flag_skips_the_first_seen_sort(oracle, vx, monkeypatch):
    """VX355_AGG_UNORDERED_OUTPUT: same groups and values, table order instead of first-seen order."""
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    rng = np.random.default_rng(63)
    n = 300000
    k = rng.integers(0, 50000, n).astype(np.int64)
    v = _dyadic(rng, n)
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, [batch_of([k, v])], [0], [abi.BIGINT], aggs, max_rows=100000)
    vx.profile_reset()
    vx.profile_enable(True)
    op = vx.Aggregation([0], [abi.BIGINT], aggs, abi.STEP_SINGLE, flags=abi.AGG_UNORDERED_OUTPUT)
    op.add_input(batch_of([k, v]))
    op.no_more_input()
    got = vx.collect_output(op, 100000)
    vx.profile_enable(False)
    assert not any("Sort" in name or "sort" in name for name in vx.profile())
    go, eo = np.argsort(got[0][0], kind="stable"), np.argsort(exp[0][0], kind="stable")
    for c in range(3):
        assert (np.asarray(got[c][0])[go] == np.asarray(exp[c][0])[eo]).all()


def _distinct_inputs(rng, n, key_shape):
    """Grouping keys of one of the table modes + a few argument / mask columns."""
    if key_shape == "array":
        keys, kinds = [rng.integers(0, 200, n).astype(np.int64)], [abi.BIGINT]
    elif key_shape == "normalized":
        keys = [(rng.integers(0, 40, n) * 1000003).astype(np.int64), rng.integers(-5, 5, n).astype(np.int32)]
        kinds = [abi.BIGINT, abi.INTEGER]
    elif key_shape == "generic":
        words = [b"alpha", b"a string that does not fit inline", b"", b"beta-beta-beta", b"z"]
        keys = [[words[i] for i in rng.integers(0, len(words), n)], rng.integers(0, 9, n).astype(np.float64) / 2]
        kinds = [abi.VARCHAR, abi.DOUBLE]
    else:
        keys, kinds = [], []
    x = rng.integers(-30, 30, n).astype(np.int64)
    d = rng.integers(0, 64, n).astype(np.float64) / 8.0  # dyadic: exact whatever the order
    r = (rng.integers(0, 32, n) / 4.0).astype(np.float32)
    m = rng.random(n) > 0.4
    return keys, kinds, x, d, r, m


@pytest.mark.parametrize("key_shape", ["array", "normalized", "generic", "global"])
@pytest.mark.parametrize("ignore_null_keys", [False, True])
def test_distinct_aggregates_next_to_plain_ones(oracle, vx, key_shape, ignore_null_keys):
    """sum / count / avg (DISTINCT x) with null arguments, null keys, a mask and plain
    aggregates in the same operator (exec/DistinctAggregations.cpp,
    GroupingSet.cpp:317-332), in every table mode, several batches, small pages."""
    rng = np.random.default_rng(77 + len(key_shape))
    n = 60000
    keys, kinds, x, d, r, m = _distinct_inputs(rng, n, key_shape)
    nk = len(keys)
    D = abi.AGG_FN_DISTINCT
    xv, dv, mv = rng.random(n) > 0.1, rng.random(n) > 0.1, rng.random(n) > 0.05
    kv = [rng.random(n) > 0.03 for _ in keys]

    def piece(lo, hi):
        cols = [abi.HostColumn(kinds[j], keys[j][lo:hi], valid=kv[j][lo:hi]) for j in range(nk)]
        cols += [abi.HostColumn(abi.BIGINT, x[lo:hi], valid=xv[lo:hi]), abi.HostColumn(abi.DOUBLE, d[lo:hi], valid=dv[lo:hi]),
                 abi.HostColumn(abi.REAL, r[lo:hi]), abi.HostColumn(abi.BOOLEAN, m[lo:hi], valid=mv[lo:hi])]
        return abi.HostBatch(cols)

    batches = [piece(i, min(n, i + 13000)) for i in range(0, n, 13000)]
    X, Dd, R, M = nk, nk + 1, nk + 2, nk + 3
    aggs = [(abi.AGG_SUM, X, abi.BIGINT, -1, -1, D), (abi.AGG_COUNT, X, abi.BIGINT, M, -1, D),
            (abi.AGG_SUM, X, abi.BIGINT), (abi.AGG_AVG, Dd, abi.DOUBLE, -1, -1, D),
            (abi.AGG_SUM, Dd, abi.DOUBLE, M, -1, D), (abi.AGG_COUNT_STAR, -1, abi.BIGINT),
            (abi.AGG_SUM, R, abi.REAL, -1, -1, D), (abi.AGG_MAX, X, abi.BIGINT, -1, -1, D),
            (abi.AGG_AVG, Dd, abi.DOUBLE)]
    kw = dict(ignore_null_keys=ignore_null_keys) if nk else {}
    exp, eop = run_agg(oracle, batches, list(range(nk)), kinds, aggs, max_rows=97, **kw)
    got, gop = run_agg(vx, batches, list(range(nk)), kinds, aggs, max_rows=97, **kw)
    assert gop.kinds == eop.kinds
    assert_columns_equal(got, exp, gop.kinds, what=f"distinct {key_shape}")


def test_distinct_only_aggregates_device_resident_many_values(oracle, vx):
    """Only DISTINCT aggregates (the operator's own table holds just the keys), device-resident
    input, 2 M rows with ~300 K distinct (group, value) pairs."""
    rng = np.random.default_rng(99)
    n = 2_000_000
    k = rng.integers(0, 3000, n).astype(np.int32)
    x = rng.integers(0, 100, n).astype(np.int64) * 7919
    hb = batch_of([k, x])
    D = abi.AGG_FN_DISTINCT
    aggs = [(abi.AGG_COUNT, 1, abi.BIGINT, -1, -1, D), (abi.AGG_SUM, 1, abi.BIGINT, -1, -1, D)]
    exp, _ = run_agg(oracle, [hb], [0], [abi.INTEGER], aggs, max_rows=1000)
    op = vx.Aggregation([0], [abi.INTEGER], aggs)
    op.add_input(vx.to_device(hb))
    op.add_input(vx.to_device(hb))  # the second pass adds no new value
    op.no_more_input()
    got = vx.collect_output(op, 1000)
    assert_columns_equal(got, exp, op.kinds, what="distinct only")


def test_distinct_aggregates_are_refused_outside_the_single_step(vx):
    with pytest.raises(Exception) as e:
        vx.Aggregation([0], [abi.BIGINT], [(abi.AGG_SUM, 1, abi.BIGINT, -1, -1, abi.AGG_FN_DISTINCT)],
                       abi.STEP_PARTIAL)
    assert "distinct inputs" in str(e.value)


_WORDS = [b"", b"a", b"a\x00", b"ab", b"abcdefg", b"abcdefgh", b"abcdefgh\x00", b"abcdefghi", b"twelve bytes", b"thirteen byte",
          b"a string that is clearly longer than twelve bytes", b"a string that is clearly longer than twelve bytes!",
          b"\xff\xfe high bytes sort last", b"\x7f", b"\x80", b"zzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzz"]


@pytest.mark.parametrize("key_shape", ["array", "normalized", "generic", "global"])
def test_min_max_over_strings(oracle, vx, key_shape):
    """min / max over VARCHAR (MinMaxAggregateBase.cpp:305-480): byte order, a string before
    its extensions, nulls skipped, a group without a value is null, masks, strings on both
    sides of the 12-byte inline limit; next to plain and DISTINCT aggregates; small pages."""
    rng = np.random.default_rng(500 + len(key_shape))
    n = 50000
    keys, kinds, x, d, r, m = _distinct_inputs(rng, n, key_shape)
    nk = len(keys)
    s = [_WORDS[i] for i in rng.integers(0, len(_WORDS), n)]
    # some groups see only nulls: null out every string of the rows whose x is a multiple of 7... and more
    sv = (rng.random(n) > 0.2) & (x % 7 != 0)
    mv = rng.random(n) > 0.05

    def piece(lo, hi):
        cols = [abi.HostColumn(kinds[j], keys[j][lo:hi]) for j in range(nk)]
        cols += [abi.HostColumn(abi.VARCHAR, s[lo:hi], valid=sv[lo:hi]), abi.HostColumn(abi.BIGINT, x[lo:hi]),
                 abi.HostColumn(abi.BOOLEAN, m[lo:hi], valid=mv[lo:hi])]
        return abi.HostBatch(cols)

    batches = [piece(i, min(n, i + 9000)) for i in range(0, n, 9000)]
    S, X, M = nk, nk + 1, nk + 2
    aggs = [(abi.AGG_MIN, S, abi.VARCHAR), (abi.AGG_MAX, S, abi.VARCHAR), (abi.AGG_SUM, X, abi.BIGINT),
            (abi.AGG_MAX, S, abi.VARCHAR, M), (abi.AGG_COUNT, X, abi.BIGINT, -1, -1, abi.AGG_FN_DISTINCT),
            (abi.AGG_MIN, S, abi.VARCHAR, M)]
    exp, eop = run_agg(oracle, batches, list(range(nk)), kinds, aggs, max_rows=61)
    got, gop = run_agg(vx, batches, list(range(nk)), kinds, aggs, max_rows=61)
    assert gop.kinds == eop.kinds
    assert_columns_equal(got, exp, gop.kinds, what=f"string min/max {key_shape}")


def test_min_max_over_strings_partial_then_final(oracle, vx):
    """The intermediate type of min / max is the input type: partial results of two operators
    merged by a FINAL one equal the SINGLE aggregation."""
    rng = np.random.default_rng(321)
    n = 30000
    k = rng.integers(0, 50, n).astype(np.int64)
    s = [_WORDS[i] for i in rng.integers(0, len(_WORDS), n)]
    sv = rng.random(n) > 0.3
    aggs = [(abi.AGG_MIN, 1, abi.VARCHAR), (abi.AGG_MAX, 1, abi.VARCHAR), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    halves = [abi.HostBatch([abi.HostColumn(abi.BIGINT, k[lo:hi]), abi.HostColumn(abi.VARCHAR, s[lo:hi], valid=sv[lo:hi])])
              for lo, hi in ((0, n // 2), (n // 2, n))]
    exp, _ = run_agg(oracle, halves, [0], [abi.BIGINT], aggs)
    partials = []
    for hb in halves:
        out, op = run_agg(vx, [hb], [0], [abi.BIGINT], aggs, abi.STEP_PARTIAL)
        assert op.kinds == [abi.BIGINT, abi.VARCHAR, abi.VARCHAR, abi.BIGINT]
        partials.append(abi.HostBatch([abi.HostColumn(kind, v if isinstance(v, list) else np.asarray(v), valid=np.asarray(valid, bool))
                                       for kind, (v, valid) in zip(op.kinds, out)]))
    final_aggs = [(abi.AGG_MIN, 1, abi.VARCHAR), (abi.AGG_MAX, 2, abi.VARCHAR), (abi.AGG_COUNT_STAR, 3, abi.BIGINT)]
    got, gop = run_agg(vx, partials, [0], [abi.BIGINT], final_aggs, abi.STEP_FINAL)
    assert_columns_equal(got, exp, gop.kinds, what="string min/max partial -> final")


def test_count_over_any_column_type(oracle, vx):
    """count(x) only reads x's validity (CountAggregate.cpp:27-147): VARCHAR (inline and not),
    TIMESTAMP, BOOLEAN, REAL inputs, flat and dictionary encoded, with and without DISTINCT."""
    rng = np.random.default_rng(4711)
    n = 40000
    k = rng.integers(0, 300, n).astype(np.int64)
    s = [_WORDS[i] for i in rng.integers(0, len(_WORDS), n)]
    ts = np.stack([rng.integers(0, 5, n), rng.integers(0, 3, n)], axis=1).astype(np.int64)
    b = rng.random(n) > 0.5
    r = rng.integers(0, 8, n).astype(np.float32)
    base = [_WORDS[i] for i in range(6)]
    cols = [abi.HostColumn(abi.BIGINT, k),
            abi.HostColumn(abi.VARCHAR, s, valid=rng.random(n) > 0.3),
            abi.HostColumn(abi.TIMESTAMP, ts, valid=rng.random(n) > 0.1),
            abi.HostColumn(abi.BOOLEAN, b, valid=rng.random(n) > 0.5),
            abi.HostColumn(abi.REAL, r),
            abi.HostColumn(abi.VARCHAR, base, valid=rng.random(n) > 0.2, encoding=abi.DICTIONARY,
                           indices=rng.integers(0, 6, n).astype(np.int32))]
    hb = abi.HostBatch(cols, n)
    D = abi.AGG_FN_DISTINCT
    aggs = [(abi.AGG_COUNT, 1, abi.VARCHAR), (abi.AGG_COUNT, 2, abi.TIMESTAMP), (abi.AGG_COUNT, 3, abi.BOOLEAN),
            (abi.AGG_COUNT, 4, abi.REAL), (abi.AGG_COUNT, 5, abi.VARCHAR), (abi.AGG_COUNT_STAR, -1, abi.BIGINT),
            (abi.AGG_COUNT, 1, abi.VARCHAR, -1, -1, D), (abi.AGG_COUNT, 3, abi.BOOLEAN, -1, -1, D)]
    for keys, kinds in (([0], [abi.BIGINT]), ([], [])):
        exp, _ = run_agg(oracle, [hb, hb], keys, kinds, aggs)
        got, gop = run_agg(vx, [hb, hb], keys, kinds, aggs)
        assert_columns_equal(got, exp, gop.kinds, what=f"count any type, keys {keys}")


@pytest.mark.parametrize("global_agg", [False, True])
def test_distinct_and_string_aggregates_on_empty_and_tiny_inputs(oracle, vx, global_agg):
    """No input at all, a zero-row batch, one row, only nulls; one row per output page."""
    D = abi.AGG_FN_DISTINCT
    keys, kinds = ([], []) if global_agg else ([0], [abi.BIGINT])
    aggs = [(abi.AGG_MIN, 1, abi.VARCHAR), (abi.AGG_SUM, 2, abi.BIGINT, -1, -1, D), (abi.AGG_COUNT, 1, abi.VARCHAR, -1, -1, D),
            (abi.AGG_COUNT_STAR, -1, abi.BIGINT), (abi.AGG_MAX, 1, abi.VARCHAR)]

    def batch(k, s, sv, x):
        return abi.HostBatch([abi.HostColumn(abi.BIGINT, np.array(k, dtype=np.int64)),
                              abi.HostColumn(abi.VARCHAR, s, valid=sv),
                              abi.HostColumn(abi.BIGINT, np.array(x, dtype=np.int64))], len(k))

    cases = {
        "no input": [],
        "zero rows": [batch([], [], [], [])],
        "one row": [batch([5], [b"only"], [True], [7])],
        "only nulls": [batch([1, 1, 2], [b"", b"", b""], [False, False, False], [3, 3, 3])],
        "one per page": [batch([3, 1, 3, 2, 1], [b"b", b"a string beyond twelve bytes", b"a", b"", b"a string beyond twelve byte"],
                               [True, True, True, True, True], [1, 2, 1, 2, 2])],
    }
    for name, batches in cases.items():
        exp, _ = run_agg(oracle, batches, keys, kinds, aggs, max_rows=1)
        got, gop = run_agg(vx, batches, keys, kinds, aggs, max_rows=1)
        assert_columns_equal(got, exp, gop.kinds, what=name)


def test_partial_flush_with_min_max_over_strings(oracle, vx):
    """A PARTIAL operator carrying min / max over VARCHAR is flushed in the middle of its input
    (HashAggregation.cpp:191-236): the flushed pages plus the rest, merged by a FINAL operator,
    equal the SINGLE aggregation; the operator starts over with empty tables after each flush."""
    rng = np.random.default_rng(8080)
    n = 24000
    k = rng.integers(0, 200, n).astype(np.int64)
    s = [_WORDS[i] for i in rng.integers(0, len(_WORDS), n)]
    sv = rng.random(n) > 0.2
    x = rng.integers(-50, 50, n).astype(np.int64)
    aggs = [(abi.AGG_MIN, 1, abi.VARCHAR), (abi.AGG_SUM, 2, abi.BIGINT), (abi.AGG_MAX, 1, abi.VARCHAR)]

    def piece(lo, hi):
        return abi.HostBatch([abi.HostColumn(abi.BIGINT, k[lo:hi]), abi.HostColumn(abi.VARCHAR, s[lo:hi], valid=sv[lo:hi]),
                              abi.HostColumn(abi.BIGINT, x[lo:hi])], hi - lo)

    exp, _ = run_agg(oracle, [piece(0, n)], [0], [abi.BIGINT], aggs)
    op = vx.Aggregation([0], [abi.BIGINT], aggs, abi.STEP_PARTIAL)
    pages = []
    for lo in range(0, n, 6000):
        op.add_input(piece(lo, lo + 6000))
        if lo in (6000, 12000):
            assert op.stats().table_bytes > 200 * 16   # the strings' set table counts towards isPartialFull
            op.flush()
            pages.append(vx.collect_output(op, 77))
            assert op.stats().num_groups == 0
    op.no_more_input()
    pages.append(vx.collect_output(op, 77))
    assert op.stats().num_flushes == 2
    partial = abi.HostBatch([abi.HostColumn(kind, sum((list(p[c][0]) for p in pages), []) if kind == abi.VARCHAR
                                            else np.concatenate([np.asarray(p[c][0]) for p in pages]),
                                            valid=np.concatenate([np.asarray(p[c][1], bool) for p in pages]))
                             for c, kind in enumerate(op.kinds)])
    final_aggs = [(abi.AGG_MIN, 1, abi.VARCHAR), (abi.AGG_SUM, 2, abi.BIGINT), (abi.AGG_MAX, 3, abi.VARCHAR)]
    got, gop = run_agg(vx, [partial], [0], [abi.BIGINT], final_aggs, abi.STEP_FINAL)
    assert_columns_equal(got, exp, gop.kinds, what="flush with string min / max")


def test_to_intermediate_passes_strings_of_min_max_through(oracle, vx):
    """Abandoned partial aggregation with min / max over VARCHAR: the value itself where the mask
    lets the row through (MinMaxAggregateBase.cpp:319-349), next to the numeric aggregates; a
    FINAL step over the result equals the SINGLE aggregation."""
    rng = np.random.default_rng(63)
    n = 30000
    k = rng.integers(0, 300, n).astype(np.int64)
    s = [_WORDS[i] for i in rng.integers(0, len(_WORDS), n)]
    svalid = rng.random(n) > 0.2
    x = _dyadic(rng, n)
    m = rng.random(n) > 0.5
    mvalid = rng.random(n) > 0.1
    b = abi.HostBatch([abi.HostColumn(abi.BIGINT, k), abi.HostColumn(abi.VARCHAR, s, valid=svalid),
                       abi.HostColumn(abi.DOUBLE, x), abi.HostColumn(abi.BOOLEAN, m, valid=mvalid)])
    raw = [(abi.AGG_MIN, 1, abi.VARCHAR), (abi.AGG_AVG, 2, abi.DOUBLE), (abi.AGG_MAX, 1, abi.VARCHAR, 3),
           (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    kinds = [abi.VARCHAR, abi.DOUBLE, abi.BIGINT, abi.VARCHAR, abi.BIGINT]
    op = vx.Aggregation([0], [abi.BIGINT], raw, abi.STEP_PARTIAL)
    cols = op.to_intermediate(b, kinds)
    ok = m & mvalid
    assert [v for v, valid in zip(cols[0][0], cols[0][1]) if valid] == [v for v, valid in zip(s, svalid) if valid]
    assert (np.asarray(cols[0][1]) == svalid).all()
    assert (np.asarray(cols[3][1]) == (svalid & ok)).all()
    assert [v for v, valid in zip(cols[3][0], cols[3][1]) if valid] == [v for v, valid in zip(s, svalid & ok) if valid]
    final_aggs = [(abi.AGG_MIN, 1, abi.VARCHAR), (abi.AGG_AVG, 2, abi.DOUBLE, -1, 3), (abi.AGG_MAX, 4, abi.VARCHAR),
                  (abi.AGG_COUNT_STAR, 5, abi.BIGINT)]
    fin = vx.Aggregation([0], [abi.BIGINT], final_aggs, abi.STEP_FINAL)
    fin.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, k)] +
                                [abi.HostColumn(kind, v if isinstance(v, list) else np.asarray(v), np.asarray(valid, bool))
                                 for kind, (v, valid) in zip(kinds, cols)]))
    fin.no_more_input()
    got = vx.collect_output(fin, 5000)
    exp_single, eop = run_agg(oracle, [b], [0], [abi.BIGINT], raw, max_rows=5000)
    assert_columns_equal(got, exp_single, eop.kinds, what="final(toIntermediate) vs single, strings")


def test_partial_rows_merge_through_presto_pages(oracle, vx):
    """The N > 1 merge with the library's own writer and reader (velox_amd/dist.py
    all_gather_partial_pages): PARTIAL on the GPU, the partial rows as checksummed PrestoPages,
    read back into HBM, FINAL - equal to the SINGLE aggregation. A one-rank stand-in replaces
    torch.distributed (the 2-rank gloo run of the same code is tests/test_dist_gloo.py)."""
    import torch
    from velox_amd import dist as vdist

    class OneRank:
        def get_world_size(self):
            return 1

        def all_gather(self, out, t):
            out[0].copy_(t)

    rng = np.random.default_rng(1717)
    n = 50000
    ids = rng.integers(0, 3000, n)
    keys = [b"a grouping key well beyond twelve bytes #%04d" % i for i in ids]
    x = rng.integers(-(1 << 40), 1 << 40, n).astype(np.int64)
    d = _dyadic(rng, n)
    b = abi.HostBatch([abi.HostColumn(abi.VARCHAR, keys), abi.HostColumn(abi.BIGINT, x, rng.random(n) > 0.05),
                       abi.HostColumn(abi.DOUBLE, d)])
    raw = [(abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_AVG, 2, abi.DOUBLE), (abi.AGG_MIN, 0, abi.VARCHAR), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    part, pop = run_agg(vx, [b], [0], [abi.VARCHAR], raw, abi.STEP_PARTIAL, max_rows=5000)
    merged = vdist.merge_partials(vx, OneRank(), torch, part, [abi.VARCHAR], raw, None)
    exp, eop = run_agg(oracle, [b], [0], [abi.VARCHAR], raw, max_rows=5000)
    assert_columns_equal(merged, exp, eop.kinds, what="merge through pages")


@pytest.mark.parametrize("num_keys,nullable", [(3, False), (4, False), (4, True)])
def test_three_and_four_grouping_keys_stay_on_the_specialised_kernel(oracle, vx, monkeypatch, num_keys, nullable):
    """BASELINE.json words configs[1] as a "4-key group-by with 6 aggregates": TPC-H Q1's two flag
    keys plus one or two low-cardinality INTEGER keys (FastShape::KX), the fused filter and
    sum / avg / count. The 4-key plan without nulls is in the ahead-of-time table (bench.py
    --workload q1x4); the others are instantiated with hiprtc. Nullable third / fourth keys use the
    null bits behind the loads (fastKeyBit)."""
    monkeypatch.setenv("VX355_JIT", "sync")
    rng = np.random.default_rng(4100 + num_keys)
    n = 250_000
    flags = [b"A", b"N", b"R"]
    rf = [flags[i] for i in rng.integers(0, 3, n)]
    ls = [bytes([c]) for c in rng.choice(list(b"FO"), n)]
    lnum = rng.integers(1, 8, n).astype(np.int32)
    mode = rng.integers(0, 7, n).astype(np.int32)
    qty = rng.integers(1, 51, n).astype(np.float64)
    ep = rng.integers(90000, 10500000, n).astype(np.float64) / 128
    disc = rng.integers(0, 11, n).astype(np.float64) / 64
    ship = rng.integers(8036, 10562, n).astype(np.int32)
    cols = [rf, ls, lnum, mode, qty, ep, disc, ship]
    valid = [None] * 8
    if nullable:
        valid[2] = rng.random(n) > 0.05
        valid[3] = rng.random(n) > 0.02
        valid[6] = rng.random(n) > 0.01
    host = batch_of(cols, valid)
    terms = [(7, abi.CMP_LE, 10471)]
    projs = [[(5, 1.0, 0.0), (6, -1.0, 1.0)]]
    keys = list(range(num_keys))
    key_types = [abi.VARCHAR, abi.VARCHAR, abi.INTEGER, abi.INTEGER][:num_keys]
    P = vx.PROJ
    aggs = [(abi.AGG_SUM, 4, abi.DOUBLE), (abi.AGG_SUM, 5, abi.DOUBLE), (abi.AGG_SUM, P(0), abi.DOUBLE),
            (abi.AGG_AVG, 4, abi.DOUBLE), (abi.AGG_AVG, 6, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    idx, proj, pnulls = oracle.filter_project(host, terms, projs, with_nulls=True)

    def take(c, v):
        picked = [c[i] for i in idx] if isinstance(c, list) else np.asarray(c)[idx]
        return picked, (None if v is None else np.asarray(v)[idx])
    picked = [take(c, v) for c, v in zip(cols, valid)]
    ref = batch_of([p[0] for p in picked[:7]] + [proj[0]], [p[1] for p in picked[:7]] + [pnulls[0]])
    ref_aggs = [(abi.AGG_SUM, 4, abi.DOUBLE), (abi.AGG_SUM, 5, abi.DOUBLE), (abi.AGG_SUM, 7, abi.DOUBLE),
                (abi.AGG_AVG, 4, abi.DOUBLE), (abi.AGG_AVG, 6, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    for ignore in ((False, True) if nullable else (False,)):
        exp, _ = run_agg(oracle, [ref], keys, key_types, ref_aggs, ignore_null_keys=ignore)
        op = vx.Aggregation(keys, key_types, aggs, ignore_null_keys=ignore)
        op.set_fused_input(terms, projs)
        vx.profile_reset()
        vx.profile_enable(True)
        op.add_input(vx.to_device(host))
        op.no_more_input()
        got = vx.collect_output(op, 500)
        vx.profile_enable(False)
        assert_columns_equal(got, exp, op.kinds, what=f"{num_keys} keys, nullable={nullable}, ignore={ignore}")
        names = vx.profile()
        assert "k_agg_fast" in names and "k_agg_lds" not in names, names
        assert len(exp[0][0]) >= 6 * 7 * (7 if num_keys == 4 else 1)


@pytest.mark.parametrize("plan", ["c1", "two_keys_int_sums_min_max", "interpreting_kernel"])
@pytest.mark.parametrize("scratch", ["1", "0"])
def test_direct_index_flush_through_scratch_copies(oracle, vx, monkeypatch, plan, scratch):
    """Many live keys in a direct-index table (BASELINE config 1: 1000 groups): the LDS kernels
    store their reduced words per workgroup and k_lds_reduce folds them into the table instead of
    one HBM atomic per (key, word, workgroup). Same results as the atomics flush
    (VX355_AGG_SCRATCH_FLUSH=0), bit for bit - counts, 128-bit BIGINT totals with carries between
    the copies, min / max, first-seen order - over several batches."""
    monkeypatch.setenv("VX355_AGG_SCRATCH_FLUSH", scratch)
    monkeypatch.setenv("VX355_JIT", "sync")
    rng = np.random.default_rng(77)
    n = 1 << 21
    if plan == "c1":
        cols = [rng.integers(0, 1000, n).astype(np.int64), _dyadic(rng, n)]
        keys, key_types = [0], [abi.BIGINT]
        aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
        valid = None
    else:
        big = rng.integers(-(1 << 40), 1 << 40, n).astype(np.int64)   # mixed signs: low words wrap, carries between copies
        cols = [rng.integers(0, 20, n).astype(np.int64), rng.integers(-5, 5, n).astype(np.int32), big,
                rng.standard_normal(n)]
        keys, key_types = [0, 1], [abi.BIGINT, abi.INTEGER]
        aggs = [(abi.AGG_SUM, 2, abi.BIGINT), (abi.AGG_MIN, 2, abi.BIGINT), (abi.AGG_MAX, 3, abi.DOUBLE),
                (abi.AGG_COUNT, 3, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
        valid = [None, None, rng.random(n) > 0.1, rng.random(n) > 0.3]
        monkeypatch.setenv("VX355_AGG_SCRATCH_MIN_ATOMICS", "1")   # 200 groups would keep the atomics flush
        if plan == "interpreting_kernel":
            monkeypatch.setenv("VX355_AGG_NO_FAST", "1")
    first = batch_of(cols, valid)
    second = batch_of([c[::-1].copy() for c in cols], None if valid is None else
                      [None if v is None else v[::-1].copy() for v in valid])
    exp, _ = run_agg(oracle, [first, second], keys, key_types, aggs, max_rows=4096)
    vx.profile_reset()
    vx.profile_enable(True)
    got, gop = run_agg(vx, [vx.to_device(first), vx.to_device(second)], keys, key_types, aggs, max_rows=4096)
    vx.profile_enable(False)
    assert_columns_equal(got, exp, gop.kinds, what=f"{plan} scratch={scratch}")
    names = vx.profile()
    assert ("k_lds_reduce" in names) == (scratch == "1"), names
    assert ("k_agg_lds" in names) == (plan == "interpreting_kernel")


@pytest.mark.parametrize("slots_only", ["1", "0"])
def test_later_batch_with_many_new_keys_overflows_the_lds_slots_and_is_replayed(oracle, vx, monkeypatch, slots_only):
    """Few groups in an open-addressing table: the LDS kernels run with a hashed slot map and the
    table is sized for workgroups x slots new groups per launch, not for a chunk of new groups
    (LdsPlan::deferOverflow). The first batch shows 100 keys; the second brings 60 000 more. Its
    workgroups run out of slots, their rows are deferred and replayed into a grown table: same
    groups, sums, counts and first-seen order as the oracle; VX355_AGG_SLOTS_ONLY=0 (tables sized
    for the chunk, overflow rows straight to HBM) gives the same answer."""
    monkeypatch.setenv("VX355_JIT", "sync")
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    monkeypatch.setenv("VX355_AGG_SLOTS_ONLY", slots_only)
    rng = np.random.default_rng(61)
    few = rng.integers(0, 1 << 40, 100).astype(np.int64)
    n1, n2 = 300_000, 1_500_000
    k1 = few[rng.integers(0, 100, n1)]
    # (the new keys lie inside the range the first batch showed: no key-range widening, whose replays
    # would count as deferred rows too)
    lo, hi = int(few.min()), int(few.max())
    k2 = np.where(rng.random(n2) < 0.06, lo + rng.integers(0, 60_000, n2).astype(np.int64) * ((hi - lo) // 60_001),
                  few[rng.integers(0, 100, n2)])
    batches = [batch_of([k1, _dyadic(rng, n1)]), batch_of([k2, _dyadic(rng, n2)])]
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, batches, [0], [abi.BIGINT], aggs, max_rows=100000)
    got, op = run_agg(vx, [vx.to_device(b) for b in batches], [0], [abi.BIGINT], aggs, max_rows=100000)
    assert_columns_equal(got, exp, op.kinds, what="slot overflow, slots_only=" + slots_only)
    st = op.stats()
    assert st.hash_mode == abi.MODE_NORMALIZED_KEY and st.num_groups == len(exp[0][0]) > 40_000
    assert (st.deferred_rows > 10_000) == (slots_only == "1")


def test_a_second_process_loads_hiprtc_instances_from_the_disk_cache(tmp_path):
    """Plan shapes outside the ahead-of-time table are compiled once per MACHINE: the code object is
    kept under VX355_CACHE_DIR, keyed by shape, architecture and a hash of the device headers. The
    first process compiles, a second process with the same cache directory loads the instance and
    compiles nothing - also with the default asynchronous mode, so a short-lived operator in a new
    worker process runs on the specialised kernel from its first batch."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = tmp_path / "worker.py"
    worker.write_text("""
import sys
import numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from velox_amd import abi, ops
from gpu_util import batch_of
ops.init(0)
rng = np.random.default_rng(5)
n = 1 << 18
hb = batch_of([rng.integers(0, 9, n).astype(np.int32), rng.integers(0, 5, n).astype(np.int64),
               rng.integers(0, 1 << 20, n) / 1024.0, rng.integers(0, 1 << 20, n) / 1024.0])
aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_MAX, 3, abi.DOUBLE), (abi.AGG_AVG, 3, abi.DOUBLE),
        (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
op = ops.Aggregation([0, 1], [abi.INTEGER, abi.BIGINT], aggs)
ops.profile_enable(True)
op.add_input(ops.to_device(hb))
op.no_more_input()
out = ops.collect_output(op, 100)
print('KERNELS', sorted(ops.profile()), 'JIT', op.stats().reserved, 'SUM', float(np.sum(out[2][0])))
""" % (root, os.path.join(root, "tests")))
    cache = tmp_path / "cache"
    cache.mkdir()
    outs = []
    for jit in ("sync", "async"):
        env = dict(os.environ, VX355_CACHE_DIR=str(cache), VX355_LOG_SHAPES="1", VX355_JIT=jit)
        r = subprocess.run([sys.executable, str(worker)], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(r)
    first, second = outs
    assert "compiling an instance" in first.stderr and "loaded from" not in first.stderr
    assert "loaded from" in second.stderr and "compiling an instance" not in second.stderr
    assert "k_agg_fast" in second.stdout and "k_agg_lds" not in second.stdout   # first batch, asynchronous mode
    assert first.stdout.split("SUM")[1] == second.stdout.split("SUM")[1]
    assert len(list(cache.glob("agg_fast_*.hsaco"))) == 1


@pytest.mark.parametrize("groups", [3000, 150_000, 6_000_000])
def test_first_seen_order_through_the_librarys_own_pair_sort(vx, groups):
    """Groups come out in first-seen order (GroupingSet.cpp:828-839). Between 4096 and 32 M groups the
    (first row, group row) pairs are put in order by the library's own LSD radix sort
    (csrc/radix_sort.hip; rocPRIM until round 5): 3000 groups take the one-workgroup kernel of the
    small tables, 150 000 the single-workgroup scan of the digit histograms, 6 M the multi-workgroup
    scan. Keys arrive in a random permutation, twice: the output must list them in the order of their
    first occurrence, with count 2 each - an order only the sort can produce (the table is indexed by key)."""
    rng = np.random.default_rng(groups)
    keys = (rng.permutation(groups).astype(np.int64) * 3 + 7)
    op = vx.Aggregation([0], [abi.BIGINT], [(abi.AGG_COUNT_STAR, -1, abi.BIGINT)])
    op.add_input(batch_of([keys]))
    op.add_input(batch_of([keys[::-1].copy()]))
    op.no_more_input()
    out = vx.collect_output(op, 1 << 20)
    got_keys, got_counts = np.asarray(out[0][0]), np.asarray(out[1][0])
    assert len(got_keys) == groups
    assert (got_keys == keys).all()
    assert (got_counts == 2).all()


@pytest.mark.parametrize("shape", ["uniform", "two_batches", "bunched_keys_overflow_a_region", "int32_key",
                                   "compact_off", "sparse", "sparse_two_batches", "sparse_one_hot_key",
                                   "sparse_compact_off"])
def test_compact_records_when_no_group_order_is_wanted(oracle, vx, shape, monkeypatch):
    """VX355_AGG_UNORDERED_OUTPUT + one flat operand: the radix passes move records without row number and
    accumulator mask. Direct-index table: 12 bytes {32-bit key | mask word, operand} instead of 16 (recLoad in
    agg.hip); open-addressing table ("sparse" shapes): 16 bytes {key, operand} instead of 24, the home slot
    recomputed from the key (hashRecLoad). Results per group equal the oracle's (compared as multisets); a
    second batch meets the groups of the first (for sparse keys: the folds that look groups up in the table
    instead of the dense ones); keys bunched into one level-1 bin / one key on half of the rows overflow an
    optimistic region, which restarts the chunk with complete records; an INTEGER key; and
    VX355_AGG_COMPACT_RECORDS=0 as the control."""
    monkeypatch.setenv("VX355_AGG_RADIX_MIN_ROWS", "1")
    monkeypatch.setenv("VX355_AGG_DENSE_MIN_ROWS", "1")
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    if shape.endswith("compact_off"):
        monkeypatch.setenv("VX355_AGG_COMPACT_RECORDS", "0")
    sparse = shape.startswith("sparse")
    rng = np.random.default_rng(515)
    # the padded range is 2 x space = 4883 partitions of 2048 groups: two levels (> 4096), and rows >= 1024 per
    # partition, the radix path's own rule
    n, space = 8_000_000, 5_000_000
    batches = []
    for b in range(2 if shape.endswith("two_batches") else 1):
        k = rng.integers(0, space, n).astype(np.int64)
        if shape == "bunched_keys_overflow_a_region":
            k[: n * 6 // 10] = rng.integers(1_000_000, 1_100_000, n * 6 // 10)  # one level-1 bin holds 131072 keys
            k = rng.permutation(k)
        if shape == "sparse_one_hot_key":
            k[rng.random(n) < 0.5] = 77
        if sparse:
            k = ((k.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(0x5DEECE66D)).astype(np.int64)
        if shape == "int32_key":
            k = k.astype(np.int32)
        batches.append(batch_of([k, _dyadic(rng, n)]))
    kind = abi.INTEGER if shape == "int32_key" else abi.BIGINT
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, batches, [0], [kind], aggs, max_rows=1 << 22)
    vx.profile_reset()
    vx.profile_enable(True)
    op = vx.Aggregation([0], [kind], aggs, abi.STEP_SINGLE, flags=abi.AGG_UNORDERED_OUTPUT)
    for b in batches:
        op.add_input(vx.to_device(b))  # (host batches arrive in staging-sized chunks: too few rows per partition)
    op.no_more_input()
    got = vx.collect_output(op, 1 << 22)
    vx.profile_enable(False)
    prof = vx.profile()
    assert "k_rp_scatter1" in prof and "k_rp_scatter2" in prof and "k_rp_aggregate" in prof
    st = op.stats()
    assert st.hash_mode == (abi.MODE_NORMALIZED_KEY if sparse else abi.MODE_ARRAY) and st.radix_launches == len(batches)
    compact = 0 if shape.endswith("compact_off") or shape in ("bunched_keys_overflow_a_region", "sparse_one_hot_key") \
        else len(batches)
    assert st.compact_record_launches == compact
    go, eo = np.argsort(got[0][0], kind="stable"), np.argsort(exp[0][0], kind="stable")
    assert len(got[0][0]) == len(exp[0][0]) > 1000
    for c in range(3):
        assert (np.asarray(got[c][0])[go] == np.asarray(exp[c][0])[eo]).all(), c
        assert np.asarray(got[c][1]).all()


@pytest.mark.parametrize("later_groups", [5, 40, 700])
@pytest.mark.parametrize("wide_plan", [False, True])
def test_first_chunk_layout_sized_from_the_first_rows_survives_a_wrong_estimate(oracle, vx, later_groups, wide_plan):
    """The first launch of a stream runs before the groups are counted: its LDS layout is sized from the distinct
    keys among the first 2048 rows (round 6). Here those rows show 3 keys and the rest of the same batch 5 / 40 / 700
    more: workgroups that run out of LDS slots update the group rows in HBM themselves, the result equals the
    oracle's (first-seen order included), with few accumulator words and with the eleven of a Q1-like plan (the
    layout with a replica per lane). GroupingSet::addInputForActiveRows semantics, exec/GroupingSet.cpp:190-365."""
    rng = np.random.default_rng(100 + later_groups + int(wide_plan))
    n = 120000
    k = np.concatenate([rng.integers(0, 3, 4096), rng.integers(0, 3 + later_groups, n - 4096)]).astype(np.int64)
    cols = [k, _dyadic(rng, n)]
    aggs = list(C1_AGGS)
    if wide_plan:
        cols += [_dyadic(rng, n), _dyadic(rng, n), _dyadic(rng, n)]
        aggs = [(abi.AGG_SUM, c, abi.DOUBLE) for c in (1, 2, 3, 4)] + \
               [(abi.AGG_AVG, c, abi.DOUBLE) for c in (1, 2)] + [(abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    batches = [batch_of(cols)]
    exp, _ = run_agg(oracle, batches, [0], [abi.BIGINT], aggs)
    got, gop = run_agg(vx, batches, [0], [abi.BIGINT], aggs)
    assert_columns_equal(got, exp, gop.kinds, what="first chunk, estimate too small")
    assert gop.stats().num_groups == len(np.unique(k))
