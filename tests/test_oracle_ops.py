"""Oracle group-by / join / filter / partition checked against independent
implementations (pandas, numpy) and the reference's hash-mode expectations
(/root/reference/velox/exec/tests/HashTableTest.cpp:607-655)."""
import math

import numpy as np
import pandas as pd
import pytest

from velox_amd import abi

M64 = (1 << 64) - 1


def _int_batch(cols, valids=None):
    hc = []
    for i, c in enumerate(cols):
        c = np.asarray(c)
        kind = {np.dtype(np.int64): abi.BIGINT, np.dtype(np.int32): abi.INTEGER,
                np.dtype(np.float64): abi.DOUBLE, np.dtype(np.float32): abi.REAL,
                np.dtype(np.int16): abi.SMALLINT, np.dtype(np.int8): abi.TINYINT}[c.dtype]
        hc.append(abi.HostColumn(kind, c, None if valids is None else valids[i]))
    return abi.HostBatch(hc)


# ---- hash mode decisions (HashTableTest.cpp testCycle) ----------------------
def _join_mode(oracle, size, ways, key_kinds, spacing=1):
    builds = []
    seq = 0
    for _ in range(ways):
        cols = []
        for k in key_kinds:
            vals = spacing * (seq + np.arange(size, dtype=np.int64))
            if k == abi.VARCHAR:
                strs = []
                for r, v in enumerate(vals):
                    s = str(int(v))
                    if r > 10000 and r % 10 == 0:
                        s += "----" + s + "----" + s
                    strs.append(s.encode())
                cols.append(abi.HostColumn(abi.VARCHAR, strs))
            else:
                cols.append(abi.HostColumn(abi.BIGINT, vals))
        b = oracle.JoinBuild(list(range(len(key_kinds))), key_kinds)
        b.add_input(abi.HostBatch(cols))
        builds.append(b)
        seq += size
    table = builds[0].finish(builds[1:])
    st = table.stats()
    assert st.num_rows == size * ways
    return st.hash_mode, table, builds


def test_mode_int2_dense_array(oracle):
    assert _join_mode(oracle, 500, 2, [abi.BIGINT, abi.BIGINT])[0] == abi.MODE_ARRAY


def test_mode_string1_dense_array(oracle):
    assert _join_mode(oracle, 500, 2, [abi.VARCHAR])[0] == abi.MODE_ARRAY


def test_mode_string2_normalized(oracle):
    assert _join_mode(oracle, 5000, 19, [abi.VARCHAR, abi.VARCHAR])[0] == abi.MODE_NORMALIZED_KEY


def test_mode_int2_sparse_array(oracle):
    assert _join_mode(oracle, 500, 2, [abi.BIGINT, abi.BIGINT], spacing=1000)[0] == abi.MODE_ARRAY


def test_mode_int2_sparse_normalized(oracle):
    assert _join_mode(oracle, 10000, 2, [abi.BIGINT, abi.BIGINT], spacing=1000)[0] == abi.MODE_NORMALIZED_KEY


def test_mode_mixed6_sparse_hash(oracle):
    kinds = [abi.BIGINT] * 5 + [abi.VARCHAR]
    assert _join_mode(oracle, 20000, 9, kinds, spacing=1000)[0] == abi.MODE_HASH


def test_probe_every_key_hits_its_row(oracle):
    """HashTableTest::testProbe (:466): every inserted key is found, others miss."""
    mode, table, _ = _join_mode(oracle, 3000, 2, [abi.BIGINT, abi.BIGINT], spacing=1000)
    keys = 1000 * np.arange(0, 7000, dtype=np.int64)
    probe = oracle.JoinProbe(table, [0, 1], abi.JOIN_LEFT)
    probe.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, keys), abi.HostColumn(abi.BIGINT, keys)]))
    mapping, build_rows, _, fin = probe.get_output(10000, build_col_ids=[])
    assert fin and list(mapping) == list(range(7000))
    assert (build_rows[:6000] == np.arange(6000)).all()
    assert (build_rows[6000:] == -1).all()


# ---- group by vs pandas --------------------------------------------------------
def _run_agg(oracle, batches, key_cols, key_types, aggs, step=abi.STEP_SINGLE, **kw):
    op = oracle.Aggregation(key_cols, key_types, aggs, step, **kw)
    for b in batches:
        op.add_input(b)
    op.no_more_input()
    return oracle.collect_output(op, 777), op


@pytest.mark.parametrize("adaptive", [True, False])
def test_groupby_c1_shape_vs_pandas(oracle, adaptive):
    rng = np.random.default_rng(7)
    n = 50000
    k = rng.integers(0, 1000, n).astype(np.int64)
    v = rng.random(n)
    batches = [_int_batch([k[i:i + 10000], v[i:i + 10000]]) for i in range(0, n, 10000)]
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT),
            (abi.AGG_MIN, 1, abi.DOUBLE), (abi.AGG_MAX, 1, abi.DOUBLE), (abi.AGG_AVG, 1, abi.DOUBLE)]
    out, op = _run_agg(oracle, batches, [0], [abi.BIGINT], aggs, hash_adaptivity=adaptive)
    assert op.stats().hash_mode == (abi.MODE_ARRAY if adaptive else abi.MODE_HASH)
    keys = out[0][0]
    # first-seen order (GroupingSet.cpp:828-839)
    _, first = np.unique(k, return_index=True)
    assert list(keys) == list(k[np.sort(first)])
    df = pd.DataFrame({"k": k, "v": v}).groupby("k", sort=False)["v"]
    ref = df.agg(["sum", "count", "min", "max", "mean"]).loc[keys]
    # Sequential sum in input order is what the reference computes; pandas may
    # reassociate, so compare sums with a few-ULP tolerance and the rest exactly.
    np.testing.assert_allclose(out[1][0], ref["sum"].values, rtol=1e-13)
    assert (out[2][0] == ref["count"].values).all()
    assert (out[3][0] == ref["min"].values).all()
    assert (out[4][0] == ref["max"].values).all()
    np.testing.assert_allclose(out[5][0], ref["mean"].values, rtol=1e-13)
    # and bit-exact against a straight Python left-to-right sum for a few groups
    for g in keys[:5]:
        s = 0.0
        for x in v[k == g]:
            s += x
        assert out[1][0][list(keys).index(g)] == s


def test_groupby_modes_and_null_keys(oracle):
    rng = np.random.default_rng(8)
    n = 20000
    # sparse keys in a huge range -> not an array; two keys -> normalized key
    k1 = (rng.integers(0, 400000, n) * 1000003).astype(np.int64)
    k2 = rng.integers(-500, 500, n).astype(np.int32)
    valid1 = rng.random(n) > 0.1
    v = rng.integers(-1000, 1000, n).astype(np.int64)
    vvalid = rng.random(n) > 0.2
    batch = abi.HostBatch([abi.HostColumn(abi.BIGINT, k1, valid1), abi.HostColumn(abi.INTEGER, k2),
                           abi.HostColumn(abi.BIGINT, v, vvalid)])
    aggs = [(abi.AGG_SUM, 2, abi.BIGINT), (abi.AGG_COUNT, 2, abi.BIGINT), (abi.AGG_AVG, 2, abi.BIGINT)]
    out, op = _run_agg(oracle, [batch], [0, 1], [abi.BIGINT, abi.INTEGER], aggs)
    assert op.stats().hash_mode == abi.MODE_NORMALIZED_KEY
    df = pd.DataFrame({"k1": pd.array(np.where(valid1, k1, 0), dtype="Int64"), "k2": k2,
                       "v": pd.array(v, dtype="Int64")})
    df.loc[~valid1, "k1"] = pd.NA
    df.loc[~vvalid, "v"] = pd.NA
    ref = df.groupby(["k1", "k2"], dropna=False, sort=False)["v"].agg(["sum", "count", "mean"])
    got = {}
    for i in range(len(out[0][0])):
        key = (int(out[0][0][i]) if out[0][1][i] else None, int(out[1][0][i]))
        got[key] = (int(out[2][0][i]) if out[2][1][i] else None, int(out[3][0][i]),
                    float(out[4][0][i]) if out[4][1][i] else None)
    assert len(got) == len(ref)
    for (a, b), row in ref.iterrows():
        key = (None if pd.isna(a) else int(a), int(b))
        s, c, m = got[key]
        assert c == row["count"]
        if row["count"] == 0:
            assert s is None and m is None  # all-null group -> NULL sum/avg, count 0
        else:
            assert s == row["sum"] and m == pytest.approx(float(row["mean"]), rel=1e-15)
    # ignoreNullKeys drops the rows with a null key
    out2, _ = _run_agg(oracle, [batch], [0, 1], [abi.BIGINT, abi.INTEGER], aggs, ignore_null_keys=True)
    assert len(out2[0][0]) == sum(1 for k in got if k[0] is not None)
    assert out2[0][1].all()


def test_groupby_string_and_double_keys_hash_mode(oracle):
    rng = np.random.default_rng(9)
    n = 5000
    flags = [bytes([c]) for c in rng.choice(list(b"ANR"), n)]
    d = rng.choice([0.0, -0.0, 1.5, float("nan"), -2.25], n)
    v = rng.random(n).astype(np.float32)
    batch = abi.HostBatch([abi.HostColumn(abi.VARCHAR, flags), abi.HostColumn(abi.DOUBLE, d),
                           abi.HostColumn(abi.REAL, v)])
    aggs = [(abi.AGG_SUM, 2, abi.REAL), (abi.AGG_MAX, 2, abi.REAL), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    out, op = _run_agg(oracle, [batch], [0, 1], [abi.VARCHAR, abi.DOUBLE], aggs)
    assert op.stats().hash_mode == abi.MODE_HASH  # DOUBLE keys have no value ids
    # 0.0 and -0.0 are one group, all NaNs are one group: 3 flags x 4 values
    assert len(out[0][0]) == 12
    total = 0
    for i in range(12):
        f, dv = out[0][0][i], out[1][0][i]
        sel = np.array([flags[j] == f and (d[j] == dv or (math.isnan(d[j]) and math.isnan(dv)))
                        for j in range(n)])
        assert out[4][0][i] == sel.sum()
        acc = 0.0
        for x in v[sel]:
            acc += float(x)  # sum(REAL) accumulates in double
        assert out[2][0][i] == np.float32(acc)
        assert out[3][0][i] == v[sel].max()
        total += sel.sum()
    assert total == n


def test_global_aggregation_and_masks(oracle):
    v = np.array([1.0, 2.0, 3.0, 4.0])
    m = abi.HostColumn(abi.BOOLEAN, [True, False, True, True], valid=[True, True, False, True])
    batch = abi.HostBatch([abi.HostColumn(abi.DOUBLE, v), m])
    aggs = [(abi.AGG_SUM, 0, abi.DOUBLE, 1), (abi.AGG_COUNT_STAR, -1, abi.BIGINT, 1),
            (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    out, _ = _run_agg(oracle, [batch], [], [], aggs)
    assert out[0][0][0] == 5.0 and out[1][0][0] == 2 and out[2][0][0] == 4
    # empty input still yields one row: NULL sum, zero counts
    out, _ = _run_agg(oracle, [], [], [], aggs)
    assert len(out[0][0]) == 1 and not out[0][1][0] and out[1][0][0] == 0


def test_sum_bigint_overflow_is_a_user_error(oracle):
    big = np.array([2 ** 62, 2 ** 62, 5], dtype=np.int64)
    op = oracle.Aggregation([], [], [(abi.AGG_SUM, 0, abi.BIGINT)])
    with pytest.raises(oracle.OracleError) as e:
        op.add_input(_int_batch([big]))
    assert e.value.status == abi.EUSER and "integer overflow" in str(e.value)


def test_partial_then_final_equals_single(oracle):
    rng = np.random.default_rng(10)
    n = 30000
    k = rng.integers(0, 97, n).astype(np.int64)
    v = rng.random(n)
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_AVG, 1, abi.DOUBLE), (abi.AGG_COUNT, 1, abi.DOUBLE),
            (abi.AGG_MIN, 1, abi.DOUBLE)]
    single, _ = _run_agg(oracle, [_int_batch([k, v])], [0], [abi.BIGINT], aggs)
    parts = []
    for lo in range(0, n, 10000):
        p, _ = _run_agg(oracle, [_int_batch([k[lo:lo + 10000], v[lo:lo + 10000]])], [0], [abi.BIGINT],
                        aggs, step=abi.STEP_PARTIAL)
        parts.append(p)
    # partial output: k, sum, (avg.sum, avg.count), count, min
    fin_aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_AVG, 2, abi.DOUBLE, -1, 3),
                (abi.AGG_COUNT, 4, abi.DOUBLE), (abi.AGG_MIN, 5, abi.DOUBLE)]
    fin = oracle.Aggregation([0], [abi.BIGINT], fin_aggs, abi.STEP_FINAL)
    for p in parts:
        cols = [abi.HostColumn(abi.BIGINT, p[0][0]), abi.HostColumn(abi.DOUBLE, p[1][0], p[1][1]),
                abi.HostColumn(abi.DOUBLE, p[2][0], p[2][1]), abi.HostColumn(abi.BIGINT, p[3][0], p[3][1]),
                abi.HostColumn(abi.BIGINT, p[4][0]), abi.HostColumn(abi.DOUBLE, p[5][0], p[5][1])]
        fin.add_input(abi.HostBatch(cols))
    fin.no_more_input()
    final = oracle.collect_output(fin)
    a = {int(kk): i for i, kk in enumerate(single[0][0])}
    for i, kk in enumerate(final[0][0]):
        j = a[int(kk)]
        assert final[1][0][i] == pytest.approx(single[1][0][j], rel=1e-14)
        assert final[2][0][i] == pytest.approx(single[2][0][j], rel=1e-14)
        assert final[3][0][i] == single[3][0][j]
        assert final[4][0][i] == single[4][0][j]


# ---- joins vs pandas -----------------------------------------------------------
@pytest.mark.parametrize("join_type", [abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_LEFT_SEMI_FILTER, abi.JOIN_ANTI])
def test_join_vs_pandas(oracle, join_type):
    rng = np.random.default_rng(11)
    nb, npb = 4000, 9000
    bk = rng.integers(0, 1500, nb).astype(np.int64)  # duplicates
    bvalid = rng.random(nb) > 0.05
    bpay = rng.integers(0, 1 << 40, nb).astype(np.int64)
    pk = rng.integers(0, 3000, npb).astype(np.int64)
    pvalid = rng.random(npb) > 0.05
    builds = []
    for lo in (0, 2500):
        b = oracle.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], join_type)
        for s in range(lo, min(nb, lo + 2500), 1000):
            e = min(lo + 2500, s + 1000, nb)
            b.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, bk[s:e], bvalid[s:e]),
                                       abi.HostColumn(abi.BIGINT, bpay[s:e])]))
        builds.append(b)
    table = builds[0].finish(builds[1:])
    st = table.stats()
    assert st.num_rows == bvalid.sum() and st.has_duplicates == 1
    assert st.num_distinct == len(np.unique(bk[bvalid]))
    probe = oracle.JoinProbe(table, [0], join_type)
    probe.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, pk, pvalid)]))
    got = []
    last = -1
    while True:
        mapping, build_rows, cols, fin = probe.get_output(1000)
        assert (np.diff(mapping) >= 0).all() and (len(mapping) == 0 or mapping[0] >= last)
        if len(mapping):
            last = mapping[-1]
        for i in range(len(mapping)):
            got.append((int(mapping[i]), int(cols[0][0][i]) if cols[0][1][i] else None))
        if fin:
            break
    bdf = pd.DataFrame({"k": bk[bvalid], "pay": bpay[bvalid]})
    pdf = pd.DataFrame({"row": np.arange(npb), "k": pk})[pvalid]
    if join_type == abi.JOIN_INNER:
        m = pdf.merge(bdf, on="k")
        want = sorted(zip(m["row"], m["pay"]))
    elif join_type == abi.JOIN_LEFT:
        m = pdf.merge(bdf, on="k", how="left")
        want = [(int(r), None if pd.isna(p) else int(p)) for r, p in zip(m["row"], m["pay"])]
        want += [(int(r), None) for r in np.arange(npb)[~pvalid]]
        want = sorted(want, key=lambda t: (t[0], -1 if t[1] is None else t[1]))
    elif join_type == abi.JOIN_LEFT_SEMI_FILTER:
        want = [(int(r), None) for r in pdf["row"][pdf["k"].isin(bdf["k"])]]
    else:
        hit = set(pdf["row"][pdf["k"].isin(bdf["k"])])
        want = [(r, None) for r in range(npb) if r not in hit]
    assert sorted(got, key=lambda t: (t[0], -1 if t[1] is None else t[1])) == list(want)


def test_filter_compact_and_partition(oracle):
    rng = np.random.default_rng(12)
    n = 1000
    v, nl, rw = rng.random(n) > 0.3, rng.random(n) > 0.1, rng.random(n) > 0.2
    assert list(oracle.filter_compact(v, nl, rw)) == list(np.nonzero(v & nl & rw)[0])
    assert list(oracle.filter_compact(v)) == list(np.nonzero(v)[0])
    xxhash = pytest.importorskip("xxhash")
    h = rng.integers(0, 1 << 63, n, dtype=np.uint64) * 2 + 1
    assert (oracle.partition(h, abi.PART_MODULO, 7) == (h % np.uint64(7))).all()
    assert (oracle.partition(h, abi.PART_BIT_RANGE, bit_begin=29, bit_end=32) == ((h >> np.uint64(29)) & np.uint64(7))).all()
    import struct
    def local(x):
        x32 = int(x) & 0xFFFFFFFF
        rev = int.from_bytes(bytes(int(f"{b:08b}"[::-1], 2) for b in x32.to_bytes(4, "little")), "little")
        return xxhash.xxh32(struct.pack("<I", rev), seed=0).intdigest()
    want = np.array([local(x) % 5 for x in h], dtype=np.uint32)
    assert (oracle.partition(h, abi.PART_LOCAL_MODULO, 5) == want).all()


def test_filter_project_vs_numpy(oracle):
    """The oracle's FilterProject restatement against straight numpy for the
    TPC-H Q1 expressions (TpchQueryBuilder.cpp:203-252)."""
    rng = np.random.default_rng(20)
    n = 5000
    ship = rng.integers(8036, 10562, n).astype(np.int32)
    ep = rng.random(n) * 1e5
    disc = rng.integers(0, 11, n) / 100.0
    tax = rng.integers(0, 9, n) / 100.0
    seg = [[b"BUILDING", b"AUTOMOBILE", b"MACHINERY"][i] for i in rng.integers(0, 3, n)]
    b = abi.HostBatch([abi.HostColumn(abi.INTEGER, ship), abi.HostColumn(abi.DOUBLE, ep),
                       abi.HostColumn(abi.DOUBLE, disc), abi.HostColumn(abi.DOUBLE, tax),
                       abi.HostColumn(abi.VARCHAR, seg)])
    idx, outs, _ = oracle.filter_project(
        b, [(0, abi.CMP_LE, 10471), (4, abi.CMP_EQ, b"BUILDING")],
        [[(1, 1.0, 0.0), (2, -1.0, 1.0)], [(1, 1.0, 0.0), (2, -1.0, 1.0), (3, 1.0, 1.0)]])
    sel = np.flatnonzero((ship <= 10471) & (np.array(seg) == b"BUILDING"))
    assert (idx == sel).all()
    assert (outs[0] == (ep * (1 - disc))[sel]).all()
    assert (outs[1] == (ep * (1 - disc) * (1 + tax))[sel]).all()


def test_min_of_all_nan_group_is_nan(oracle):
    """MinMaxAggregateBase.cpp:293-303: min starts at quiet_NaN, so a group
    holding only NaNs keeps NaN; NaN and inf gives inf."""
    k = np.array([1, 1, 2, 2, 3], dtype=np.int64)
    v = np.array([np.nan, np.nan, np.nan, np.inf, 1.0])
    out, _ = _run_agg(oracle, [_int_batch([k, v])], [0], [abi.BIGINT],
                      [(abi.AGG_MIN, 1, abi.DOUBLE), (abi.AGG_MAX, 1, abi.DOUBLE)])
    assert math.isnan(out[1][0][0]) and out[1][0][1] == np.inf and out[1][0][2] == 1.0
    assert math.isnan(out[2][0][0]) and math.isnan(out[2][0][1]) and out[2][0][2] == 1.0


def test_split_block_bloom_filter_known_answers_and_properties(oracle):
    """SplitBlockBloomFilterTest.cpp: numBlocks known answers (:42-51, both SIMD widths),
    no false negatives and < 3 % false positives for contiguous and random values with
    folly::hasher (:53-126)."""
    assert oracle.bloom_num_blocks(50_000_000, 0.01, 8) * 32 == 60509568
    assert oracle.bloom_num_blocks(50_000_000, 0.01, 4) * 16 == 65766912
    assert oracle.bloom_num_blocks(45_523_964, 0.1, 8) * 32 == 32848640
    assert oracle.bloom_num_blocks(45_523_964, 0.1, 4) * 16 == 27546352
    rng = np.random.default_rng(42)
    size = 100_000
    for lanes in (8, 4):
        members = np.flatnonzero(rng.integers(0, 10, size) == 0).astype(np.int64)
        blocks = oracle.bloom_build(members, lanes)
        got = oracle.bloom_test(blocks, np.arange(size, dtype=np.int64))
        is_member = np.zeros(size, dtype=bool)
        is_member[members] = True
        assert got[is_member].all()
        assert got[~is_member].mean() < 0.03
        values = rng.integers(-2 ** 63, 2 ** 63 - 1, size, dtype=np.int64)
        blocks = oracle.bloom_build(values, lanes)
        assert oracle.bloom_test(blocks, values).all()
        others = rng.integers(-2 ** 63, 2 ** 63 - 1, size, dtype=np.int64)
        assert oracle.bloom_test(blocks, others).mean() < 0.03


def _join_case(seed=5, nb=400, npb=900):
    rng = np.random.default_rng(seed)
    bk = rng.integers(0, 120, nb).astype(np.int64)
    bvalid = rng.random(nb) > 0.1
    bpay = np.arange(nb, dtype=np.int64) * 10
    pk = rng.integers(-20, 160, npb).astype(np.int64)
    pvalid = rng.random(npb) > 0.1
    return bk, bvalid, bpay, pk, pvalid


def _oracle_join(oracle, join_type, bk, bvalid, bpay, pk, pvalid, null_aware=False, max_rows=97):
    b1 = oracle.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], join_type, null_aware)
    b2 = oracle.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], join_type, null_aware)
    h = len(bk) // 2
    b1.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, bk[:h], bvalid[:h]), abi.HostColumn(abi.BIGINT, bpay[:h])]))
    b2.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, bk[h:], bvalid[h:]), abi.HostColumn(abi.BIGINT, bpay[h:])]))
    table = b1.finish([b2])
    probe = oracle.JoinProbe(table, [0], join_type, null_aware)
    probe.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, pk, pvalid)]))
    pairs = []
    while True:
        mapping, rows, cols, fin = probe.get_output(max_rows)
        pairs += [(int(m), None if not cols[0][1][i] else int(cols[0][0][i])) for i, m in enumerate(mapping)]
        if fin:
            break
    build_side = []
    if join_type in (abi.JOIN_RIGHT, abi.JOIN_FULL, abi.JOIN_RIGHT_SEMI_FILTER):
        while True:
            rows, cols, fin = probe.get_build_side_output(max_rows)
            build_side += [int(cols[0][0][i]) for i in range(len(rows))]
            if fin:
                break
    return pairs, build_side, (b1, b2, table)


def test_oracle_right_full_semi_joins_against_brute_force(oracle):
    """core/PlanNode.h:3081-3165 semantics restated in the oracle (HashProbe.cpp:993-1080,
    1274-1377), checked against nested loops in Python: right / full list unmatched build rows
    (null keys included) after the probe, right semi filter lists the matched build rows once,
    left semi project lists every probe row with its match flag."""
    bk, bvalid, bpay, pk, pvalid = _join_case()
    match = {}
    for j in range(len(bk)):
        if bvalid[j]:
            match.setdefault(int(bk[j]), []).append(int(bpay[j]))
    inner = sorted((i, p) for i in range(len(pk)) if pvalid[i] for p in match.get(int(pk[i]), []))
    hit_pays = {p for i in range(len(pk)) if pvalid[i] for p in match.get(int(pk[i]), [])}
    kept = [int(bpay[j]) for j in range(len(bk)) if bvalid[j]]
    # right: inner pairs + every build row (null keys too) that no probe row matched
    pairs, build_side, _ = _oracle_join(oracle, abi.JOIN_RIGHT, bk, bvalid, bpay, pk, pvalid)
    assert sorted(pairs) == inner
    assert build_side == [int(p) for p in bpay if int(p) not in hit_pays]
    # full: left pairs + the same build rows
    pairs, build_side, _ = _oracle_join(oracle, abi.JOIN_FULL, bk, bvalid, bpay, pk, pvalid)
    misses = [(i, None) for i in range(len(pk)) if not (pvalid[i] and int(pk[i]) in match)]
    assert sorted(pairs, key=lambda t: (t[0], -1 if t[1] is None else t[1])) == sorted(
        inner + misses, key=lambda t: (t[0], -1 if t[1] is None else t[1]))
    assert build_side == [int(p) for p in bpay if int(p) not in hit_pays]
    # right semi filter: matched build rows, once, in build order; no probe-side output
    pairs, build_side, _ = _oracle_join(oracle, abi.JOIN_RIGHT_SEMI_FILTER, bk, bvalid, bpay, pk, pvalid)
    assert pairs == [] and build_side == [p for p in kept if p in hit_pays]
    # left semi project: every probe row once
    pairs, _, _ = _oracle_join(oracle, abi.JOIN_LEFT_SEMI_PROJECT, bk, bvalid, bpay, pk, pvalid)
    assert [m for m, _ in pairs] == list(range(len(pk)))


def test_oracle_null_aware_anti_join(oracle):
    """NOT IN semantics (HashProbe.cpp:1316-1328, HashBuild's antiJoinHasNullKeys)."""
    bk, bvalid, bpay, pk, pvalid = _join_case(seed=6)
    keys = {int(bk[j]) for j in range(len(bk)) if bvalid[j]}
    # build side with a null key: nothing qualifies
    pairs, _, _ = _oracle_join(oracle, abi.JOIN_ANTI, bk, bvalid, bpay, pk, pvalid, null_aware=True)
    assert pairs == []
    # no nulls on the build side: non-null probe keys without a match
    allv = np.ones(len(bk), dtype=bool)
    keys_all = {int(k) for k in bk}
    pairs, _, _ = _oracle_join(oracle, abi.JOIN_ANTI, bk, allv, bpay, pk, pvalid, null_aware=True)
    assert [m for m, _ in pairs] == [i for i in range(len(pk)) if pvalid[i] and int(pk[i]) not in keys_all]
    # empty build side: every probe row, null keys included
    pairs, _, _ = _oracle_join(oracle, abi.JOIN_ANTI, bk[:0], allv[:0], bpay[:0], pk, pvalid, null_aware=True)
    assert [m for m, _ in pairs] == list(range(len(pk)))
    # not null aware for comparison: rows without a match, null keys included
    pairs, _, _ = _oracle_join(oracle, abi.JOIN_ANTI, bk, bvalid, bpay, pk, pvalid)
    assert [m for m, _ in pairs] == [i for i in range(len(pk)) if not (pvalid[i] and int(pk[i]) in keys)]


def test_sequential_double_sum_is_within_the_reference_tolerance_of_the_exact_sum(oracle):
    """The reference adds DOUBLE inputs in row order (SumAggregateBase.h:48-175 via
    AggregationHook.h:126-135); its tests accept a relative difference of 2 * FLT_EPSILON
    or an absolute one below kEpsilon = 1e-5 (type/Variant.cpp:1027-1050, Variant.h:611).
    The restatement's sequential sum must lie inside that tolerance of the exact sum
    (math.fsum) — the GPU side is held to the much tighter 1 ULP of the same exact sum
    (tests/test_gpu_double_sums.py), so the two agree within the reference's own bar."""
    import math
    rng = np.random.default_rng(77)
    n = 400000
    k = rng.integers(0, 40, n).astype(np.int64)
    v = rng.integers(90000, 10500000, n) / 100.0
    op = oracle.Aggregation([0], [abi.BIGINT], [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_AVG, 1, abi.DOUBLE)])
    op.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, k), abi.HostColumn(abi.DOUBLE, v)]))
    op.no_more_input()
    out = oracle.collect_output(op, 1024)
    flt_eps = float(np.finfo(np.float32).eps)
    worst_ulp = 0.0
    for key, s, a in zip(out[0][0], out[1][0], out[2][0]):
        sel = v[k == key]
        exact = math.fsum(sel)
        # bit-identical to the row-order sum of the same doubles
        seq = 0.0
        for x in sel.tolist():
            seq += x
        assert s == seq
        assert abs(s - exact) <= max(abs(s), abs(exact)) * 2 * flt_eps
        assert abs(a - exact / len(sel)) <= abs(a) * 2 * flt_eps
        worst_ulp = max(worst_ulp, abs(s - exact) / np.spacing(exact))
    # the sequential sum is NOT within 1 ULP of the exact sum on this data (10 000 rows per group):
    # bit equality with any parallel order is not attainable, closeness to the exact sum is
    assert worst_ulp > 1


def _brute_force_join(bk, bv, pk, pv, bvalid=None, pvalid=None, flt=None):
    """Nested loops: for every probe row the list of build rows with an equal non-null key that
    pass flt(probe value, build value)."""
    out = []
    for i in range(len(pk)):
        m = []
        if pvalid is None or pvalid[i]:
            for j in range(len(bk)):
                if (bvalid is None or bvalid[j]) and bk[j] == pk[i] and (flt is None or flt(pv[i], bv[j])):
                    m.append(j)
        out.append(m)
    return out


JOIN_FILTER_KINDS = [abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_RIGHT, abi.JOIN_FULL, abi.JOIN_LEFT_SEMI_FILTER,
                     abi.JOIN_LEFT_SEMI_PROJECT, abi.JOIN_ANTI, abi.JOIN_RIGHT_SEMI_FILTER,
                     abi.JOIN_RIGHT_SEMI_PROJECT, abi.JOIN_RIGHT_ANTI]


@pytest.mark.parametrize("join_type", JOIN_FILTER_KINDS)
@pytest.mark.parametrize("with_filter", [False, True])
def test_oracle_joins_with_extra_filter_against_nested_loops(oracle, join_type, with_filter):
    """HashProbe::evalFilter semantics per join kind (HashProbe.cpp:1487-1713) and the build-side
    outputs of the right-side kinds, against nested loops: filter = probe.v < build.w."""
    rng = np.random.default_rng(41 + join_type)
    nb, npb = 300, 500
    bk = rng.integers(0, 60, nb).astype(np.int64)
    bvalid = rng.random(nb) > 0.1
    bw = rng.integers(0, 100, nb).astype(np.int64)
    bwvalid = rng.random(nb) > 0.1
    pk = rng.integers(-5, 70, npb).astype(np.int64)
    pvalid = rng.random(npb) > 0.1
    pv = rng.integers(0, 100, npb).astype(np.int64)
    pvvalid = rng.random(npb) > 0.1
    b = oracle.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], join_type)
    b.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, bk, bvalid), abi.HostColumn(abi.BIGINT, bw, bwvalid)]))
    t = b.finish()
    p = oracle.JoinProbe(t, [0], join_type)
    if with_filter:
        p.set_filter([(("probe", 1), abi.CMP_LT, ("build", 0))])
    p.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, pk, pvalid), abi.HostColumn(abi.BIGINT, pv, pvvalid)]))
    pairs = []
    while True:
        m, r, cols, fin = p.get_output(37, [0])
        pairs += list(zip(m.tolist(), r.tolist()))
        if fin:
            break
    keeps_null_rows = join_type in (abi.JOIN_RIGHT, abi.JOIN_FULL, abi.JOIN_RIGHT_SEMI_PROJECT, abi.JOIN_RIGHT_ANTI)
    kept = [j for j in range(nb) if bvalid[j] or keeps_null_rows]          # build row id -> input row
    rid = {j: i for i, j in enumerate(kept)}

    def flt(pvi, bwj):
        return True
    matches = []
    for i in range(npb):
        m = []
        if pvalid[i]:
            for j in range(nb):
                if bvalid[j] and bk[j] == pk[i] and (not with_filter or (pvvalid[i] and bwvalid[j] and pv[i] < bw[j])):
                    m.append(rid[j])
        matches.append(m)
    exp = []
    for i, m in enumerate(matches):
        if join_type in (abi.JOIN_INNER, abi.JOIN_RIGHT):
            exp += [(i, j) for j in m]
        elif join_type in (abi.JOIN_LEFT, abi.JOIN_FULL):
            exp += [(i, j) for j in m] if m else [(i, -1)]
        elif join_type == abi.JOIN_LEFT_SEMI_FILTER and m:
            exp.append((i, -1))
        elif join_type == abi.JOIN_ANTI and not m:
            exp.append((i, -1))
        elif join_type == abi.JOIN_LEFT_SEMI_PROJECT:
            exp.append((i, 0 if m else -1))
    got = pairs
    if join_type == abi.JOIN_LEFT_SEMI_PROJECT:
        got = [(i, 0 if j >= 0 else -1) for i, j in pairs]          # which match is reported is chain order
    assert [i for i, _ in got] == [i for i, _ in exp]                  # ascending probe rows, same multiplicity
    assert sorted(got) == sorted(exp)
    probed = set(j for m in matches for j in m)
    if join_type in (abi.JOIN_RIGHT, abi.JOIN_FULL, abi.JOIN_RIGHT_ANTI, abi.JOIN_RIGHT_SEMI_FILTER,
                     abi.JOIN_RIGHT_SEMI_PROJECT):
        ids = [0, abi.BUILD_COL_MATCH] if join_type == abi.JOIN_RIGHT_SEMI_PROJECT else [0]
        rows, flags = [], []
        while True:
            r, cols, fin = p.get_build_side_output(41, ids)
            rows += r.tolist()
            if join_type == abi.JOIN_RIGHT_SEMI_PROJECT:
                flags += np.asarray(cols[1][0]).tolist()
            if fin:
                break
        if join_type == abi.JOIN_RIGHT_SEMI_FILTER:
            assert rows == sorted(probed)
        elif join_type == abi.JOIN_RIGHT_SEMI_PROJECT:
            assert rows == list(range(len(kept))) and [bool(f) for f in flags] == [r in probed for r in rows]
        else:
            assert rows == [r for r in range(len(kept)) if r not in probed]


@pytest.mark.parametrize("join_type", [abi.JOIN_COUNTING_LEFT_SEMI_FILTER, abi.JOIN_COUNTING_ANTI])
def test_oracle_counting_joins_are_intersect_all_and_except_all(oracle, join_type):
    """core/PlanNode.h:3112-3116,3152-3156: INTERSECT ALL / EXCEPT ALL over multisets, across batches."""
    rng = np.random.default_rng(51)
    bk = rng.integers(0, 30, 200).astype(np.int64)
    pk = rng.integers(-3, 35, 400).astype(np.int64)
    pvalid = rng.random(400) > 0.05
    b = oracle.JoinBuild([0], [abi.BIGINT], [], [], join_type)
    b.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, bk)]))
    p = oracle.JoinProbe(b.finish(), [0], join_type)
    left = {}
    for k in bk.tolist():
        left[k] = left.get(k, 0) + 1
    got, exp = [], []
    for lo in range(0, 400, 150):
        hi = min(400, lo + 150)
        p.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, pk[lo:hi], pvalid[lo:hi])]))
        while True:
            m, r, cols, fin = p.get_output(33, [])
            got += (m + lo).tolist()
            if fin:
                break
        for i in range(lo, hi):
            consumed = bool(pvalid[i]) and left.get(int(pk[i]), 0) > 0
            if consumed:
                left[int(pk[i])] -= 1
            if consumed == (join_type == abi.JOIN_COUNTING_LEFT_SEMI_FILTER):
                exp.append(i)
    assert got == exp


def test_distinct_aggregates_vs_pandas(oracle):
    """sum / count / avg (DISTINCT x) with nulls, a mask and plain aggregates beside
    them (exec/DistinctAggregations.cpp; GroupingSet.cpp:317-332): every group keeps
    the set of its input values; pandas drop_duplicates is the independent reference."""
    rng = np.random.default_rng(4242)
    n = 40000
    k = rng.integers(0, 300, n).astype(np.int64)
    x = rng.integers(-20, 20, n).astype(np.int64)        # many repeats per group
    d = rng.integers(0, 16, n).astype(np.float64) / 4.0  # dyadic: sums exact in any order
    xv = rng.random(n) > 0.1
    m = rng.random(n) > 0.5
    mv = rng.random(n) > 0.05
    D = abi.AGG_FN_DISTINCT
    def piece(lo, hi):
        return abi.HostBatch([abi.HostColumn(abi.BIGINT, k[lo:hi]), abi.HostColumn(abi.BIGINT, x[lo:hi], valid=xv[lo:hi]),
                              abi.HostColumn(abi.DOUBLE, d[lo:hi]),
                              abi.HostColumn(abi.BOOLEAN, m[lo:hi], valid=mv[lo:hi])])
    batches = [piece(i, min(n, i + 7000)) for i in range(0, n, 7000)]
    aggs = [(abi.AGG_SUM, 1, abi.BIGINT, -1, -1, D), (abi.AGG_COUNT, 1, abi.BIGINT, -1, -1, D),
            (abi.AGG_AVG, 2, abi.DOUBLE, -1, -1, D), (abi.AGG_SUM, 2, abi.DOUBLE, 3, -1, D),
            (abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_MIN, 1, abi.BIGINT, -1, -1, D), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    out, _ = _run_agg(oracle, batches, [0], [abi.BIGINT], aggs)
    keys = out[0][0]
    _, first = np.unique(k, return_index=True)
    assert list(keys) == list(k[np.sort(first)])
    df = pd.DataFrame({"k": k, "x": np.where(xv, x, 0), "xv": xv, "d": d, "m": m & mv})
    for pos, g in enumerate(keys):
        rows = df[df.k == g]
        xs = rows[rows.xv].x.drop_duplicates()
        assert bool(out[1][1][pos]) == (len(xs) > 0)
        if len(xs):
            assert out[1][0][pos] == xs.sum()
        assert out[2][0][pos] == len(xs)
        ds = rows.d.drop_duplicates()
        assert out[3][0][pos] == ds.sum() / len(ds)
        dm = rows[rows.m].d.drop_duplicates()
        assert bool(out[4][1][pos]) == (len(dm) > 0)
        if len(dm):
            assert out[4][0][pos] == dm.sum()
        assert out[5][0][pos] == rows[rows.xv].x.sum() or not out[5][1][pos]
        if rows.xv.any():
            assert out[6][0][pos] == rows[rows.xv].x.min()
        assert out[7][0][pos] == len(rows)


def test_distinct_aggregates_are_refused_outside_the_single_step(oracle):
    with pytest.raises(oracle.OracleError) as e:
        oracle.Aggregation([0], [abi.BIGINT], [(abi.AGG_SUM, 1, abi.BIGINT, -1, -1, abi.AGG_FN_DISTINCT)],
                           step=abi.STEP_PARTIAL)
    assert e.value.status == abi.EUSER and "distinct inputs" in str(e.value)


def test_distinct_doubles_treat_every_nan_as_one_value_and_zero_signs_as_equal(oracle):
    nan2 = np.frombuffer(np.array([0x7ff8000000000001], dtype=np.uint64).tobytes(), dtype=np.float64)[0]
    v = np.array([np.nan, nan2, 0.0, -0.0, 1.5, 1.5])
    out, _ = _run_agg(oracle, [_int_batch([v])], [], [], [(abi.AGG_COUNT, 0, abi.DOUBLE, -1, -1, abi.AGG_FN_DISTINCT)])
    assert out[0][0][0] == 3


def test_min_max_over_strings_vs_python(oracle):
    """Non-numeric min / max (MinMaxAggregateBase.cpp:305-480): bytes compared as unsigned,
    a prefix before its extensions, nulls skipped, no value -> null."""
    words = [b"", b"a", b"a\x00", b"ab", b"\x80", b"\x7f", b"longer than twelve bytes", b"longer than twelve bytes!"]
    rng = np.random.default_rng(11)
    n = 5000
    k = rng.integers(0, 40, n).astype(np.int64)
    s = [words[i] for i in rng.integers(0, len(words), n)]
    valid = (rng.random(n) > 0.3) & (k % 5 != 0)
    batch = abi.HostBatch([abi.HostColumn(abi.BIGINT, k), abi.HostColumn(abi.VARCHAR, s, valid=valid)])
    out, _ = _run_agg(oracle, [batch], [0], [abi.BIGINT], [(abi.AGG_MIN, 1, abi.VARCHAR), (abi.AGG_MAX, 1, abi.VARCHAR)])
    for pos, g in enumerate(out[0][0]):
        vals = [s[i] for i in range(n) if k[i] == g and valid[i]]
        assert bool(out[1][1][pos]) == bool(vals) == bool(out[2][1][pos])
        if vals:
            assert out[1][0][pos] == min(vals) and out[2][0][pos] == max(vals)


def _i32(v):
    return int(v).to_bytes(4, "little", signed=True)


def test_presto_page_known_answers(oracle):
    """Pages assembled by hand from the format description: 21-byte header (rows, codec,
    uncompressed size, size, checksum), column count, then per column the encoding name, the
    row count, [end offsets,] the null flag, [MSB-first null bits,] [byte count,] non-null values
    (VectorStream.cpp:207-299, PrestoSerializerSerializationUtils.h:37-45)."""
    b = abi.HostBatch([abi.HostColumn(abi.BIGINT, np.array([1, 0, 3], dtype=np.int64), valid=[True, False, True]),
                       abi.HostColumn(abi.VARCHAR, [b"ab", b"", b""], valid=[True, False, True]),
                       abi.HostColumn(abi.BOOLEAN, [True, False, True])])
    (page,) = oracle.presto_serialize(b, [0, 3])
    body = _i32(3)
    body += _i32(10) + b"LONG_ARRAY" + _i32(3) + b"\x01" + bytes([0b01000000]) + (1).to_bytes(8, "little") + (3).to_bytes(8, "little")
    body += _i32(14) + b"VARIABLE_WIDTH" + _i32(3) + _i32(2) + _i32(2) + _i32(2) + b"\x01" + bytes([0b01000000]) + _i32(2) + b"ab"
    body += _i32(10) + b"BYTE_ARRAY" + _i32(3) + b"\x00" + b"\x01\x00\x01"
    want = _i32(3) + b"\x00" + _i32(len(body)) + _i32(len(body)) + bytes(8) + body
    assert page == want
    # with a listener: codec bit 4 and crc32(data | codec | rows | size), zlib's polynomial
    import zlib
    (summed,) = oracle.presto_serialize(b, [0, 3], flags=abi.PAGE_CHECKSUM)
    assert summed[4] == 4 and summed[21:] == body
    crc = zlib.crc32(body + b"\x04" + _i32(3) + _i32(len(body))) & 0xffffffff
    assert summed[13:21] == crc.to_bytes(8, "little")


def test_presto_page_row_column_known_answer(oracle):
    """A struct column assembled by hand from VectorStream::flush's ROW branch
    (serializers/VectorStream.cpp:236-262) and serializeRowVector
    (PrestoSerializerSerializationUtils.cpp:883-919): "ROW", the number of fields, the fields' streams -
    each holding only the rows of the NON-NULL structs -, the row count, rows + 1 offsets (0, then + 1
    behind every non-null struct), the null flag and the MSB-first null bits. The intermediate type of
    avg, ROW(DOUBLE sum, BIGINT count) (AverageAggregateBase.h:66-260), next to a key column."""
    sums = np.array([1.5, 0.0, -2.25, 8.0])
    counts = np.array([3, 0, 1, 5], dtype=np.int64)
    row = abi.HostRowColumn([abi.HostColumn(abi.DOUBLE, sums), abi.HostColumn(abi.BIGINT, counts, valid=[True, True, True, False])],
                            valid=[True, False, True, True])
    b = abi.HostBatch([abi.HostColumn(abi.INTEGER, np.array([7, 8, 9, 10], dtype=np.int32)), row])
    (page,) = oracle.presto_serialize(b, [0, 4])
    f64 = lambda v: np.float64(v).tobytes()  # noqa: E731
    body = _i32(2)
    body += _i32(9) + b"INT_ARRAY" + _i32(4) + b"\x00" + b"".join(_i32(v) for v in (7, 8, 9, 10))
    body += _i32(3) + b"ROW" + _i32(2)
    # the fields see structs 0, 2, 3 only
    body += _i32(10) + b"LONG_ARRAY" + _i32(3) + b"\x00" + f64(1.5) + f64(-2.25) + f64(8.0)
    body += _i32(10) + b"LONG_ARRAY" + _i32(3) + b"\x01" + bytes([0b00100000]) + (3).to_bytes(8, "little") + (1).to_bytes(8, "little")
    body += _i32(4) + b"".join(_i32(v) for v in (0, 1, 1, 2, 3)) + b"\x01" + bytes([0b01000000])
    want = _i32(4) + b"\x00" + _i32(len(body)) + _i32(len(body)) + bytes(8) + body
    assert page == want
    # a struct column without null structs: no bitmap, the fields hold every row
    dense = abi.HostBatch([abi.HostRowColumn([abi.HostColumn(abi.BIGINT, counts)])])
    (page,) = oracle.presto_serialize(dense, [1, 3])
    body = _i32(1) + _i32(3) + b"ROW" + _i32(1)
    body += _i32(10) + b"LONG_ARRAY" + _i32(2) + b"\x00" + (0).to_bytes(8, "little") + (1).to_bytes(8, "little")
    body += _i32(2) + _i32(0) + _i32(1) + _i32(2) + b"\x00"
    assert page == _i32(2) + b"\x00" + _i32(len(body)) + _i32(len(body)) + bytes(8) + body


from presto_page_reader import check_pages_decode_to_rows as _check_pages_decode_to_rows, millis as _millis, random_page_batch as _random_page_batch, random_row_page_batch as _random_row_page_batch  # noqa: E402


def test_presto_pages_with_struct_columns_decode_back_to_the_rows(oracle):
    rng = np.random.default_rng(616)
    n = 2500
    batch, py, kinds = _random_row_page_batch(rng, n)
    rows = rng.permutation(n).astype(np.int32)
    offsets = [0, 0, 1, 9, 2057, 2057, n]
    for flags in (0, abi.PAGE_CHECKSUM):
        pages = oracle.presto_serialize(batch, offsets, rows, flags)
        _check_pages_decode_to_rows(pages, py, kinds, offsets, rows)


def test_presto_pages_decode_back_to_the_rows(oracle):
    rng = np.random.default_rng(606)
    n = 3000
    batch, py = _random_page_batch(rng, n)
    kinds = [c.kind for c in batch.columns]
    rows = rng.permutation(n).astype(np.int32)
    offsets = [0, 0, 1, 9, 2057, 2057, n]
    for flags in (0, abi.PAGE_CHECKSUM):
        pages = oracle.presto_serialize(batch, offsets, rows, flags)
        _check_pages_decode_to_rows(pages, _millis(py), kinds, offsets, rows)
    pages = oracle.presto_serialize(batch, offsets, rows, abi.PAGE_LOSSLESS_TIMESTAMP)
    _check_pages_decode_to_rows(pages, py, kinds, offsets, rows, lossless=True)
    # without a row list the ranges address the batch rows themselves
    pages = oracle.presto_serialize(batch, [0, 100, n])
    _check_pages_decode_to_rows(pages, _millis(py), kinds, [0, 100, n], None)


def test_count_of_a_string_column_counts_its_non_null_rows(oracle):
    s = [b"a", b"", b"a string longer than twelve bytes", b"a", b"a string longer than twelve byteS", b"abcdX", b"abcdY"]
    batch = abi.HostBatch([abi.HostColumn(abi.VARCHAR, s, valid=[True, False, True, True, True, True, True])])
    out, _ = _run_agg(oracle, [batch], [], [], [(abi.AGG_COUNT, 0, abi.VARCHAR),
                                               (abi.AGG_COUNT, 0, abi.VARCHAR, -1, -1, abi.AGG_FN_DISTINCT)])
    assert out[0][0][0] == 6 and out[1][0][0] == 5


def _nav_join(impl, join_type, bcols, bvalids, pcols, pvalids, kinds, max_rows=333):
    """Join with nullAsValue over len(kinds) keys; build payload = build row number."""
    nk = len(kinds)
    b = impl.JoinBuild(list(range(nk)), kinds, [] if join_type in (abi.JOIN_COUNTING_LEFT_SEMI_FILTER, abi.JOIN_COUNTING_ANTI) else [nk],
                       [] if join_type in (abi.JOIN_COUNTING_LEFT_SEMI_FILTER, abi.JOIN_COUNTING_ANTI) else [abi.BIGINT],
                       join_type, False, True)
    nb = len(bcols[0])
    cols = [abi.HostColumn(kinds[k], bcols[k], bvalids[k]) for k in range(nk)]
    cols.append(abi.HostColumn(abi.BIGINT, np.arange(nb, dtype=np.int64)))
    b.add_input(abi.HostBatch(cols, nb))
    table = b.finish()
    probe = impl.JoinProbe(table, list(range(nk)), join_type, False, True)
    probe.add_input(abi.HostBatch([abi.HostColumn(kinds[k], pcols[k], pvalids[k]) for k in range(nk)], len(pcols[0])))
    out = []
    while True:
        mapping, rows, cols, fin = probe.get_output(max_rows)
        for i, m in enumerate(mapping):
            out.append((int(m), int(rows[i])))
        if fin:
            break
    return out


def _nav_cases(rng, nb, npb, kinds):
    def column(kind, n):
        if kind == abi.VARCHAR:
            words = [b"", b"a", b"bb", b"a key of more than twelve bytes"]
            return [words[i] for i in rng.integers(0, len(words), n)]
        return rng.integers(0, 6, n).astype(np.int64)
    bcols = [column(k, nb) for k in kinds]
    pcols = [column(k, npb) for k in kinds]
    bvalids = [rng.random(nb) > 0.3 for _ in kinds]
    pvalids = [rng.random(npb) > 0.3 for _ in kinds]
    def keyof(cols, valids, r):
        return tuple((cols[k][r] if not isinstance(cols[k][r], np.integer) else int(cols[k][r])) if valids[k][r] else None
                     for k in range(len(kinds)))
    bkeys = [keyof(bcols, bvalids, r) for r in range(nb)]
    pkeys = [keyof(pcols, pvalids, r) for r in range(npb)]
    return bcols, bvalids, pcols, pvalids, bkeys, pkeys


@pytest.mark.parametrize("kinds", [[abi.BIGINT], [abi.BIGINT, abi.BIGINT], [abi.VARCHAR, abi.BIGINT]])
def test_null_as_value_joins_against_nested_loops(oracle, kinds):
    """HashJoinNode::isNullAsValue (core/PlanNode.h:3442-3445): keys compare IS NOT DISTINCT FROM.
    Inner / left pairs and the counting joins (INTERSECT ALL / EXCEPT ALL) against plain Python
    over key tuples in which None equals None."""
    rng = np.random.default_rng(909 + len(kinds))
    nb, npb = 300, 700
    bcols, bvalids, pcols, pvalids, bkeys, pkeys = _nav_cases(rng, nb, npb, kinds)
    assert any(None in k for k in bkeys) and any(None in k for k in pkeys)
    want_inner = sorted((i, j) for i in range(npb) for j in range(nb) if pkeys[i] == bkeys[j])
    got = _nav_join(oracle, abi.JOIN_INNER, bcols, bvalids, pcols, pvalids, kinds)
    assert sorted(got) == want_inner
    got = _nav_join(oracle, abi.JOIN_LEFT, bcols, bvalids, pcols, pvalids, kinds)
    matched = {i for i, _ in want_inner}
    assert sorted(got) == sorted(want_inner + [(i, -1) for i in range(npb) if i not in matched])
    # counting joins: every probe row consumes one occurrence of its key while any is left
    from collections import Counter
    left = Counter(bkeys)
    inter, exc = [], []
    for i, k in enumerate(pkeys):
        if left[k] > 0:
            left[k] -= 1
            inter.append(i)
        else:
            exc.append(i)
    assert [m for m, _ in _nav_join(oracle, abi.JOIN_COUNTING_LEFT_SEMI_FILTER, bcols, bvalids, pcols, pvalids, kinds)] == inter
    assert [m for m, _ in _nav_join(oracle, abi.JOIN_COUNTING_ANTI, bcols, bvalids, pcols, pvalids, kinds)] == exc


@pytest.mark.parametrize("build", ["regular", "with_null_key", "empty", "only_null_keys"])
def test_oracle_null_aware_left_semi_project_truth_table(oracle, build):
    """x IN (subquery) as a three-valued column (HashProbe::fillLeftSemiProjectMatchColumn,
    HashProbe.cpp:923-966): build_rows_out = first match (TRUE), -1 (FALSE), -2 (NULL), against the
    SQL truth table written out in Python."""
    bk = {"regular": [1, 2, 2, 5], "with_null_key": [1, 2, None, 5], "empty": [], "only_null_keys": [None, None]}[build]
    pk = [1, 3, None, 5, 2, None, 9]
    b = oracle.JoinBuild([0], [abi.BIGINT], [], [], abi.JOIN_LEFT_SEMI_PROJECT, True)
    b.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, np.array([0 if v is None else v for v in bk], dtype=np.int64),
                                              valid=np.array([v is not None for v in bk], dtype=bool))], len(bk)))
    probe = oracle.JoinProbe(b.finish(), [0], abi.JOIN_LEFT_SEMI_PROJECT, True)
    probe.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, np.array([0 if v is None else v for v in pk], dtype=np.int64),
                                                  valid=np.array([v is not None for v in pk], dtype=bool))]))
    mapping, rows, _, fin = probe.get_output(100, [])
    assert fin and list(mapping) == list(range(len(pk)))
    values = [v for v in bk if v is not None]
    has_null = any(v is None for v in bk)
    for x, r in zip(pk, rows):
        if not bk:
            want = "F"                       # IN over an empty set
        elif x is not None and x in values:
            want = "T"
        elif x is None or has_null:
            want = "N"                       # unknown: a NULL on either side and no definite match
        else:
            want = "F"
        assert ("T" if r >= 0 else "F" if r == -1 else "N") == want, (x, r)


def test_sum_bigint_overflow_rules(oracle):
    """The oracle's two sum(BIGINT) rules: the reference's checkedPlus on the running sum in input
    order (vector/AggregationHook.h:126-135) and the order-independent 'exact total must fit int64'
    rule libvx355 implements. They differ exactly where a prefix overflows and comes back."""
    big = 1 << 62
    aggs = [(abi.AGG_SUM, 1, abi.BIGINT)]

    def run(values, rule):
        oracle.set_sum_overflow_rule(rule)
        try:
            op = oracle.Aggregation([0], [abi.BIGINT], aggs)
            op.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, np.zeros(len(values), dtype=np.int64)),
                                        abi.HostColumn(abi.BIGINT, np.array(values, dtype=np.int64))]))
            op.no_more_input()
            return int(np.asarray(oracle.collect_output(op, 16)[1][0])[0])
        finally:
            oracle.set_sum_overflow_rule(oracle.SUM_RULE_REFERENCE)

    for values in ([big, -big, big, -big], [big, big - 1], [-big, -big], [1, 2, 3]):
        assert run(values, oracle.SUM_RULE_REFERENCE) == run(values, oracle.SUM_RULE_TOTAL) == sum(values)
    # a prefix reaches 2^63, the total is 0: only the reference's rule throws
    with pytest.raises(oracle.OracleError, match="integer overflow"):
        run([big, big, -big, -big], oracle.SUM_RULE_REFERENCE)
    assert run([big, big, -big, -big], oracle.SUM_RULE_TOTAL) == 0
    assert run([-big, -big, -big, big, big], oracle.SUM_RULE_TOTAL) == -big
    # the total itself does not fit: both throw
    for values in ([big, big], [big, big, big, -big], [-big, -big, -1]):
        for rule in (oracle.SUM_RULE_REFERENCE, oracle.SUM_RULE_TOTAL):
            with pytest.raises(oracle.OracleError, match="integer overflow"):
                run(values, rule)


def test_double_comparisons_order_nan_like_the_reference(oracle):
    """functions/prestosql/Comparisons.h:42-121 (util::floating_point::NaNAware*): NaN = NaN, and
    NaN is greater than every other DOUBLE, +inf included — in FilterProject's terms."""
    v = np.array([1.0, np.nan, np.inf, -np.inf, 0.0, np.nan])
    b = abi.HostBatch([abi.HostColumn(abi.DOUBLE, v)])
    nan = float("nan")

    def sel(cmp, const):
        return oracle.filter_project(b, [(0, cmp, const)], [])[0].tolist()
    assert sel(abi.CMP_EQ, nan) == [1, 5] and sel(abi.CMP_NE, nan) == [0, 2, 3, 4]
    assert sel(abi.CMP_LT, nan) == [0, 2, 3, 4] and sel(abi.CMP_LE, nan) == [0, 1, 2, 3, 4, 5]
    assert sel(abi.CMP_GT, nan) == [] and sel(abi.CMP_GE, nan) == [1, 5]
    assert sel(abi.CMP_GT, np.inf) == [1, 5] and sel(abi.CMP_GE, np.inf) == [1, 2, 5]
    assert sel(abi.CMP_LT, 1.0) == [3, 4] and sel(abi.CMP_LE, 1.0) == [0, 3, 4]


def sql_in_with_filter(pk, pvalid, pv, pvvalid, bk, bvalid, bw, bwvalid):
    """Three-valued `x IN (SELECT y FROM build WHERE probe.v < build.w)` per probe row, straight from
    SQL: TRUE if some filtered y equals x; else FALSE if no build row passes the filter; else NULL if
    x is null or some filtered y is null; else FALSE. -> list of True / False / None."""
    out = []
    for i in range(len(pk)):
        passing = [j for j in range(len(bk)) if pvvalid[i] and bwvalid[j] and pv[i] < bw[j]]
        if pvalid[i] and any(bvalid[j] and bk[j] == pk[i] for j in passing):
            out.append(True)
        elif not passing:
            out.append(False)
        elif not pvalid[i] or any(not bvalid[j] for j in passing):
            out.append(None)
        else:
            out.append(False)
    return out


def null_aware_filter_case(seed, nb=120, npb=400, build_nulls=0.1):
    rng = np.random.default_rng(seed)
    bk = rng.integers(0, 40, nb).astype(np.int64)
    bvalid = rng.random(nb) >= build_nulls
    bw = rng.integers(0, 100, nb).astype(np.int64)
    bwvalid = rng.random(nb) > 0.1
    pk = rng.integers(-5, 50, npb).astype(np.int64)
    pvalid = rng.random(npb) > 0.1
    pv = rng.integers(0, 110, npb).astype(np.int64)     # some values pass no build row at all
    pvvalid = rng.random(npb) > 0.1
    return bk, bvalid, bw, bwvalid, pk, pvalid, pv, pvvalid


def run_null_aware_filter_join(impl, join_type, case, max_rows=53):
    bk, bvalid, bw, bwvalid, pk, pvalid, pv, pvvalid = case
    h = len(bk) // 2
    builds = []
    for lo, hi in ((0, h), (h, len(bk))):
        b = impl.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], join_type, True)
        b.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, bk[lo:hi], bvalid[lo:hi]),
                                   abi.HostColumn(abi.BIGINT, bw[lo:hi], bwvalid[lo:hi])]))
        builds.append(b)
    table = builds[0].finish(builds[1:])
    p = impl.JoinProbe(table, [0], join_type, True)
    p.set_filter([(("probe", 1), abi.CMP_LT, ("build", 0))])
    p.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, pk, pvalid), abi.HostColumn(abi.BIGINT, pv, pvvalid)]))
    pairs = []
    while True:
        m, r, cols, fin = p.get_output(max_rows, [])
        pairs += list(zip(m.tolist(), r.tolist()))
        if fin:
            break
    return pairs


@pytest.mark.parametrize("build_nulls", [0.0, 0.1, 1.0])
@pytest.mark.parametrize("join_type", [abi.JOIN_ANTI, abi.JOIN_LEFT_SEMI_PROJECT])
def test_oracle_null_aware_joins_with_extra_filter_follow_sql(oracle, join_type, build_nulls):
    """HashProbe::evalFilterForNullAwareJoin (HashProbe.cpp:1639-1700) restated in the oracle against
    SQL's three-valued IN / NOT IN over the filtered subquery: null-key probe rows meet every build
    row, unmatched rows meet the null-key build rows."""
    case = null_aware_filter_case(77 + join_type, build_nulls=build_nulls)
    bk, bvalid, bw, bwvalid, pk, pvalid, pv, pvvalid = case
    truth = sql_in_with_filter(pk, pvalid, pv, pvvalid, bk, bvalid, bw, bwvalid)
    pairs = run_null_aware_filter_join(oracle, join_type, case)
    if join_type == abi.JOIN_ANTI:
        assert [i for i, _ in pairs] == [i for i, t in enumerate(truth) if t is False]   # NOT IN is TRUE
    else:
        assert [i for i, _ in pairs] == list(range(len(pk)))
        got = [True if r >= 0 else (None if r == -2 else False) for _, r in pairs]
        assert got == truth
    assert {True, False, None} <= set(truth) or build_nulls in (0.0, 1.0)


@pytest.mark.parametrize("shape", ["normalized_unique", "normalized_duplicates", "hash_strings"])
def test_parallel_join_build_equals_the_serial_build(oracle, shape):
    """HashTable::parallelJoinBuild restated (oracle/table.h; exec/HashTable.cpp:1003-1203: rows
    partitioned by bucket range, one inserter per partition, overflow rows serially) - what bench.py's
    multi-thread CPU leg of the joins times - must build the same join as the serial insert: same
    statistics, same (probe row, payload) pairs for every probe row, duplicates as sets."""
    rng = np.random.default_rng(11)
    ways, per = 6, 40_000
    keys = rng.permutation(1 << 26)[: ways * per].astype(np.int64) * 1021      # sparse: normalized-key mode
    if shape == "normalized_duplicates":
        keys[::3] = keys[1::3][: len(keys[::3])]
    pay = np.arange(ways * per, dtype=np.int64)
    probe = np.concatenate([keys[rng.integers(0, len(keys), 200_000)], rng.integers(0, 1 << 40, 50_000).astype(np.int64)])
    if shape == "hash_strings":
        as_bytes = lambda a: [b"key-%018d" % v for v in a]     # 22 bytes: generic hash mode
        build_cols = lambda sl: abi.HostBatch([abi.HostColumn(abi.VARCHAR, as_bytes(keys[sl])), abi.HostColumn(abi.BIGINT, pay[sl])])
        probe_batch = abi.HostBatch([abi.HostColumn(abi.VARCHAR, as_bytes(probe))])
        kind = abi.VARCHAR
    else:
        build_cols = lambda sl: _int_batch([keys[sl], pay[sl]])
        probe_batch = _int_batch([probe])
        kind = abi.BIGINT
    results = []
    for threads in (1, 4):
        oracle.set_join_build_threads(threads)
        try:
            builds = []
            for w in range(ways):
                b = oracle.JoinBuild([0], [kind], [1], [abi.BIGINT], abi.JOIN_INNER)
                b.add_input(build_cols(slice(w * per, (w + 1) * per)))
                builds.append(b)
            table = builds[0].finish(builds[1:])
        finally:
            oracle.set_join_build_threads(1)
        st = table.stats()
        p = oracle.JoinProbe(table, [0], abi.JOIN_INNER)
        p.add_input(probe_batch)
        pairs = []
        while True:
            m, r, cols, fin = p.get_output(1 << 18, [0])
            pairs += list(zip(np.asarray(m).tolist(), np.asarray(cols[0][0]).tolist()))
            if fin:
                break
        results.append(((st.num_rows, st.num_distinct, st.capacity, st.hash_mode, bool(st.has_duplicates)), sorted(pairs)))
    assert results[0][0] == results[1][0]
    assert results[0][0][3] == (0 if shape == "hash_strings" else 2)
    assert results[0][1] == results[1][1] and len(results[0][1]) >= 200_000
