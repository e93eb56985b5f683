"""Parity of the standalone HIP kernels with the CPU oracle, through the C ABI:
VectorHasher::hash, value ids (range mode), filter compaction, partitioning.
Bit-exact: everything here is integer work."""
import numpy as np
import pytest

from velox_amd import abi

pytestmark = pytest.mark.gpu


def _col(kind, values, valid=None, **kw):
    return abi.HostColumn(kind, values, valid, **kw)


def _strings(rng, n, max_len):
    out = []
    for _ in range(n):
        ln = int(rng.integers(0, max_len + 1))
        out.append(bytes(rng.integers(32, 127, ln).astype(np.uint8)))
    return out


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 4097, 100003])
def test_hash_bigint_sizes_and_selection(oracle, vx, n):
    rng = np.random.default_rng(n)
    v = rng.integers(-2 ** 63, 2 ** 63 - 1, n, dtype=np.int64)
    if n > 3:
        v[:3] = [0, np.iinfo(np.int64).min, np.iinfo(np.int64).max]
    valid = rng.random(n) > 0.2
    rows = rng.random(n) > 0.3
    b = abi.HostBatch([_col(abi.BIGINT, v, valid)], n)
    init = rng.integers(0, 2 ** 63, max(1, n)).astype(np.uint64)
    exp = oracle.hash_columns(b, [0], rows, out=init.copy())
    got = vx.hash_columns(b, [0], rows, out=init.copy())
    assert (exp == got).all()
    # all rows, no selection
    assert (oracle.hash_columns(b, [0]) == vx.hash_columns(b, [0])).all()


def test_hash_every_type_multi_key_and_mix(oracle, vx):
    rng = np.random.default_rng(5)
    n = 20011
    cols = [
        _col(abi.BOOLEAN, rng.random(n) > 0.5, rng.random(n) > 0.1),
        _col(abi.TINYINT, rng.integers(-128, 128, n).astype(np.int8)),
        _col(abi.SMALLINT, rng.integers(-2 ** 15, 2 ** 15, n).astype(np.int16), rng.random(n) > 0.1),
        _col(abi.INTEGER, rng.integers(-2 ** 31, 2 ** 31, n).astype(np.int32)),
        _col(abi.BIGINT, rng.integers(-2 ** 62, 2 ** 62, n).astype(np.int64)),
        _col(abi.REAL, rng.choice([0.0, -0.0, 1.5, np.nan, -np.inf, 3.25e10], n).astype(np.float32)),
        _col(abi.DOUBLE, rng.choice([0.0, -0.0, 2.5, np.nan, np.inf, -1e300, 5e-324], n)),
        _col(abi.VARCHAR, _strings(rng, n, 40), rng.random(n) > 0.05),
        _col(abi.TIMESTAMP, np.stack([rng.integers(-10 ** 9, 10 ** 9, n),
                                      rng.integers(0, 10 ** 9, n)], axis=1)),
    ]
    b = abi.HostBatch(cols, n)
    for k in range(len(cols)):
        assert (oracle.hash_columns(b, [k]) == vx.hash_columns(b, [k])).all(), k
    keys = [4, 7, 0, 6, 3]
    assert (oracle.hash_columns(b, keys) == vx.hash_columns(b, keys)).all()
    seed = rng.integers(0, 2 ** 63, n).astype(np.uint64)
    assert (oracle.hash_columns(b, [1, 5], mix_first=True, out=seed.copy()) ==
            vx.hash_columns(b, [1, 5], mix_first=True, out=seed.copy())).all()


def test_hash_string_all_lengths(oracle, vx):
    strs = [bytes((i * 7 + j) % 251 + 1 for j in range(i)) for i in range(0, 80)]
    b = abi.HostBatch([_col(abi.VARCHAR, strs)], len(strs))
    assert (oracle.hash_columns(b, [0]) == vx.hash_columns(b, [0])).all()


def test_hash_dictionary_constant_and_device_memory(oracle, vx):
    rng = np.random.default_rng(6)
    n = 5000
    base = rng.integers(-1000, 1000, 37).astype(np.int64)
    idx = rng.integers(0, 37, n).astype(np.int32)
    dict_col = _col(abi.BIGINT, base, rng.random(n) > 0.1, encoding=abi.DICTIONARY, indices=idx)
    const_col = _col(abi.DOUBLE, np.array([2.75]), encoding=abi.CONSTANT)
    null_const = _col(abi.INTEGER, np.array([7], dtype=np.int32), np.array([False]),
                      encoding=abi.CONSTANT)
    b = abi.HostBatch([dict_col, const_col, null_const], n)
    exp = oracle.hash_columns(b, [0, 1, 2])
    assert (exp == vx.hash_columns(b, [0, 1, 2])).all()
    # same columns resident in HBM, output in HBM
    db = vx.to_device(b)
    out = vx.DeviceArray(n, np.uint64).zero()
    import ctypes as C
    st = vx.lib().vx355_hash_columns(db.ref(), abi.i32_array([0, 1, 2]), 3, None, 0, out.ptr,
                                     abi.MEM_DEVICE)
    assert st == abi.OK
    assert (out.to_host() == exp).all()


@pytest.mark.parametrize("lookup", [False, True])
def test_value_ids_range_mode(oracle, vx, lookup):
    rng = np.random.default_rng(12)
    n = 30001
    a = rng.integers(-50, 60, n).astype(np.int64)      # some outside [-40, 40]
    b_ = rng.integers(0, 10, n).astype(np.int32)
    s = [bytes([c]) if c else b"" for c in rng.choice([0, 65, 78, 82, 90], n)]
    flag = rng.random(n) > 0.5
    va, vs = rng.random(n) > 0.1, rng.random(n) > 0.1
    batch = abi.HostBatch([_col(abi.BIGINT, a, va), _col(abi.INTEGER, b_), _col(abi.VARCHAR, s, vs),
                           _col(abi.BOOLEAN, flag)], n)
    lo_s, hi_s = 0, (ord("R") + 256)
    specs = [(-40, 40, 1), (0, 9, 82), (lo_s, hi_s, 82 * 11), (0, 1, 82 * 11 * (hi_s - lo_s + 2))]
    rows = rng.random(n) > 0.25
    e_res, e_rows, e_mapped = oracle.value_ids(batch, [0, 1, 2, 3], specs, rows, lookup)
    g_res, g_rows, g_mapped = vx.value_ids(batch, [0, 1, 2, 3], specs, rows, lookup)
    if lookup:
        assert (e_rows == g_rows).all()
        sel = e_rows
    else:
        assert e_mapped == g_mapped and not g_mapped
        # rows the oracle could map must agree
        ok = rows & ~(va & ((a < -40) | (a > 40))) & ~(vs & np.array([len(x) == 1 and x[0] > 82 for x in s]))
        sel = ok
    assert (e_res[sel] == g_res[sel]).all()
    # everything in range -> all mapped
    specs2 = [(-50, 59, 1), (0, 9, 200)]
    e = oracle.value_ids(batch, [0, 1], specs2)
    g = vx.value_ids(batch, [0, 1], specs2)
    assert e[2] and g[2] and (e[0] == g[0]).all()


def test_value_ids_type_extremes(oracle, vx):
    v = np.array([np.iinfo(np.int64).min, -1, 0, 1, np.iinfo(np.int64).max], dtype=np.int64)
    batch = abi.HostBatch([_col(abi.BIGINT, v)], len(v))
    for spec in [(np.iinfo(np.int64).min, np.iinfo(np.int64).min + 10, 1),
                 (np.iinfo(np.int64).max - 10, np.iinfo(np.int64).max, 1), (-1, 1, 3)]:
        e = oracle.value_ids(batch, [0], [spec], lookup=True)
        g = vx.value_ids(batch, [0], [spec], lookup=True)
        assert (e[1] == g[1]).all() and (e[0][e[1]] == g[0][g[1]]).all()


@pytest.mark.parametrize("n", [0, 1, 64, 4095, 4096, 4097, 300007])
@pytest.mark.parametrize("density", [0.0, 0.03, 0.5, 0.985, 1.0])
def test_filter_compact(oracle, vx, n, density):
    rng = np.random.default_rng(n + int(density * 1000))
    values = rng.random(n) < density
    nulls = rng.random(n) > 0.1
    rows = rng.random(n) > 0.1
    for args in [(values,), (values, nulls), (values, nulls, rows), (values, None, rows)]:
        e = oracle.filter_compact(*args)
        g = vx.filter_compact(*args)
        assert len(e) == len(g) and (e == g).all()


def test_filter_compact_device_resident(oracle, vx):
    rng = np.random.default_rng(99)
    n = 1 << 20
    values = rng.random(n) < 0.985
    bits = vx.DeviceArray(abi.pack_bits(values))
    out = vx.DeviceArray(n, np.int32)
    cnt = vx.filter_compact_device(bits.ptr, n, out.ptr)
    assert cnt == values.sum()
    assert (out.to_host(cnt) == np.flatnonzero(values)).all()


@pytest.mark.parametrize("kind,kw", [
    (abi.PART_MODULO, dict(num_partitions=8)), (abi.PART_MODULO, dict(num_partitions=7)),
    (abi.PART_MODULO, dict(num_partitions=1000003)),
    (abi.PART_BIT_RANGE, dict(bit_begin=61, bit_end=64)), (abi.PART_BIT_RANGE, dict(bit_begin=29, bit_end=32)),
    (abi.PART_LOCAL_MODULO, dict(num_partitions=13)),
    (abi.PART_LOCAL_BIT_RANGE, dict(bit_begin=3, bit_end=9))])
def test_partition(oracle, vx, kind, kw):
    rng = np.random.default_rng(3)
    h = rng.integers(0, 2 ** 64, 100001, dtype=np.uint64)
    h[:4] = [0, 1, 2 ** 64 - 1, 2 ** 63]
    assert (oracle.partition(h, kind, **kw) == vx.partition(h, kind, **kw)).all()
    assert len(vx.partition(h[:0], kind, **kw)) == 0


def test_filter_project_q1_q3_expressions(oracle, vx):
    rng = np.random.default_rng(20)
    n = 200003
    ship = rng.integers(8036, 10562, n).astype(np.int32)
    ep = rng.random(n) * 1e5
    disc = rng.integers(0, 11, n) / 100.0
    tax = rng.integers(0, 9, n) / 100.0
    qty = rng.integers(1, 51, n).astype(np.int64)
    seg = [[b"BUILDING", b"AUTOMOBILE", b"MACHINERY", b"", b"BUILDINGS"][i] for i in rng.integers(0, 5, n)]
    valid_ship, valid_disc = rng.random(n) > 0.02, rng.random(n) > 0.02
    b = abi.HostBatch([_col(abi.INTEGER, ship, valid_ship), _col(abi.DOUBLE, ep),
                       _col(abi.DOUBLE, disc, valid_disc), _col(abi.DOUBLE, tax),
                       _col(abi.VARCHAR, seg), _col(abi.BIGINT, qty)])
    q1_proj = [[(1, 1.0, 0.0), (2, -1.0, 1.0)], [(1, 1.0, 0.0), (2, -1.0, 1.0), (3, 1.0, 1.0)],
               [(5, 2.0, 0.5), (-1, 0.0, 3.0)]]
    cases = [
        ([(0, abi.CMP_LE, 10471)], q1_proj),
        ([(0, abi.CMP_GT, 9204), (4, abi.CMP_EQ, b"BUILDING")], q1_proj[:1]),
        ([(4, abi.CMP_NE, b"BUILDING"), (1, abi.CMP_LT, 5e4), (5, abi.CMP_GE, 10)], []),
        ([], q1_proj[:2]),
        ([(0, abi.CMP_EQ, 1)], q1_proj[:1]),  # nothing passes
    ]
    for terms, projs in cases:
        e_idx, e_out, e_nulls = oracle.filter_project(b, terms, projs, with_nulls=True)
        g_idx, g_out, g_nulls = vx.filter_project(b, terms, projs, with_nulls=True)
        assert len(e_idx) == len(g_idx) and (e_idx == g_idx).all()
        for j in range(len(projs)):
            assert (e_nulls[j] == g_nulls[j]).all()
            assert (e_out[j][e_nulls[j]] == g_out[j][g_nulls[j]]).all()  # bit-exact doubles


def test_filter_terms_order_nan_like_the_reference(oracle, vx):
    """DOUBLE comparisons are NaN aware (functions/prestosql/Comparisons.h:42-121): NaN = NaN, NaN
    above +inf. Every operator against NaN, inf and finite constants, GPU == oracle."""
    rng = np.random.default_rng(23)
    pool = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1.5, -2.25, 1e300])
    v = pool[rng.integers(0, len(pool), 20011)]
    b = abi.HostBatch([_col(abi.DOUBLE, v)])
    for const in (float("nan"), float("inf"), 1.5, -0.0):
        for cmp in (abi.CMP_EQ, abi.CMP_NE, abi.CMP_LT, abi.CMP_LE, abi.CMP_GT, abi.CMP_GE):
            e_idx = oracle.filter_project(b, [(0, cmp, const)], [])[0]
            g_idx = vx.filter_project(b, [(0, cmp, const)], [])[0]
            assert len(e_idx) == len(g_idx) and (e_idx == g_idx).all(), (const, cmp)


@pytest.mark.parametrize("num_parts", [1, 2, 8, 13, 64])
@pytest.mark.parametrize("n", [0, 1, 4095, 4097, 100003])
def test_partition_scatter_is_a_stable_partition(vx, num_parts, n):
    rng = np.random.default_rng(n * 100 + num_parts)
    parts = rng.integers(0, num_parts, n).astype(np.uint32)
    k = rng.integers(-2 ** 62, 2 ** 62, n).astype(np.int64)
    d = rng.integers(0, 1 << 20, n).astype(np.int32)
    s16 = rng.integers(0, 255, (n, 16)).astype(np.uint8)
    b = rng.integers(0, 255, n).astype(np.uint8)
    outs, counts = vx.partition_scatter(parts, num_parts, [k, d, s16, b])
    order = np.argsort(parts, kind="stable")
    assert (counts == np.bincount(parts, minlength=num_parts)).all()
    for got, col in zip(outs, [k, d, s16, b]):
        assert (got == col[order]).all()


@pytest.mark.parametrize("kind", ["bigint", "integer", "string"])
def test_distinct_value_id_mode_matches_the_vector_hasher(oracle, vx, kind):
    """VectorHasher::computeValueIds / lookupValueIds in distinct-value mode (VectorHasher.cpp:128-161,
    196-224,408-492; valueId, VectorHasher.h:567-580): ids are the insertion numbers of the distinct
    values in row order, across batches, with selection bitmaps, nulls and a multiplier; once the set
    reaches its range size the batch is reported unmappable."""
    rng = np.random.default_rng(17)
    n = 50000
    pool = rng.integers(-10**12, 10**12, 3000)
    if kind == "bigint":
        mk = lambda idx: abi.HostColumn(abi.BIGINT, pool[idx].astype(np.int64), rng.random(len(idx)) > 0.05)
        k = abi.BIGINT
    elif kind == "integer":
        mk = lambda idx: abi.HostColumn(abi.INTEGER, (pool[idx] % 100000).astype(np.int32), rng.random(len(idx)) > 0.05)
        k = abi.INTEGER
    else:
        words = [b"w%05d" % i for i in range(3000)] + [b""]
        mk = lambda idx: abi.HostColumn(abi.VARCHAR, [words[i] for i in idx.tolist()], rng.random(len(idx)) > 0.05)
        k = abi.VARCHAR
    batches = [mk(rng.integers(0, 1000, n)), mk(rng.integers(0, 3000, n)), mk(rng.integers(500, 2500, 777))]
    sels = [None, rng.random(n) > 0.3, None]
    h = oracle.Hasher(k)
    for col, sel in zip(batches, sels):                       # analysis pass fills the distinct set
        h.compute_value_ids(col, rows=np.ones(col.num_rows, bool) if sel is None else sel)
    for mult in (1, 7):
        h = oracle.Hasher(k)
        h.compute_value_ids(batches[0], rows=np.ones(n, bool))
        range_size = h.enable_value_ids(mult, 50) // mult      # = rangeSize_
        d = vx.ValueDict(k, range_size)
        for col, sel in zip(batches, sels):
            rows = np.ones(col.num_rows, bool) if sel is None else sel
            base = rng.integers(0, 5, col.num_rows).astype(np.uint64)
            ok_e, ids_e = h.compute_value_ids(col, rows=rows, result=base.copy())
            ok_g, ids_g = d.compute(abi.HostBatch([col]), 0, rows=sel, multiplier=mult, result=base.copy())
            assert ok_g == ok_e
            assert d.size() == h.state().num_distinct
            if ok_e:
                assert (ids_g[rows] == ids_e[rows]).all()
        probe = mk(rng.integers(0, 3000, 5000))
        prow = rng.random(5000) > 0.2
        rows_e, ids_e = h.lookup_value_ids(probe, prow.copy())
        rows_g, ids_g = d.lookup(abi.HostBatch([probe]), 0, rows=prow, multiplier=mult)
        assert (np.asarray(rows_g) == np.asarray(rows_e)).all()
        assert (ids_g[np.asarray(rows_e)] == ids_e[np.asarray(rows_e)]).all()


@pytest.mark.parametrize("device_resident", [False, True])
def test_presto_pages_equal_the_oracle_byte_for_byte(oracle, vx, device_resident):
    """vx355_presto_serialize vs oracle/presto_page.h: every kind, null bitmaps present and
    absent, a row list, page sizes around the 8-row null byte and the 2048-row tile, empty
    ranges, strings beyond the inline limit, checksums, lossless timestamps; and the pages
    decode back to the input rows through the independent reader."""
    from presto_page_reader import check_pages_decode_to_rows, millis, random_page_batch
    rng = np.random.default_rng(707)
    n = 9000
    batch, py = random_page_batch(rng, n)
    kinds = [c.kind for c in batch.columns]
    rows = rng.permutation(n).astype(np.int32)
    offsets = [0, 0, 1, 8, 17, 2065, 4113, 4113, 4114 + 2047, n]
    src = vx.to_device(batch) if device_resident else batch
    for flags in (0, abi.PAGE_CHECKSUM, abi.PAGE_LOSSLESS_TIMESTAMP):
        exp = oracle.presto_serialize(batch, offsets, rows, flags)
        got = vx.presto_serialize(src, offsets, rows, flags)
        assert [len(g) for g in got] == [len(e) for e in exp]
        assert got == exp, f"flags {flags}"
    check_pages_decode_to_rows(vx.presto_serialize(src, offsets, rows), millis(py), kinds, offsets, rows)
    # pages left in HBM (for an exchange that sends from device memory); no checksum there
    assert vx.presto_serialize(src, offsets, rows, device_out=True) == oracle.presto_serialize(batch, offsets, rows)
    with pytest.raises(Exception) as e:
        vx.presto_serialize(src, offsets, rows, abi.PAGE_CHECKSUM, device_out=True)
    assert "host output" in str(e.value)
    # no row list: the ranges are batch rows; a batch without any null has no bitmap at all
    dense, _ = random_page_batch(rng, 5000, with_nulls=False)
    assert vx.presto_serialize(dense, [0, 2048, 5000]) == oracle.presto_serialize(dense, [0, 2048, 5000])


def test_presto_pages_of_encoded_columns_are_flattened(oracle, vx):
    """Dictionary and constant inputs (the iterative serializer always flattens,
    serializers/PrestoSerializer.h:27-35)."""
    rng = np.random.default_rng(808)
    n = 5000
    base = rng.integers(0, 1000, 50).astype(np.int64)
    idx = rng.integers(0, 50, n).astype(np.int32)
    words = [b"alpha", b"a dictionary value longer than twelve bytes", b""]
    cols = [abi.HostColumn(abi.BIGINT, base, valid=rng.random(n) > 0.2, encoding=abi.DICTIONARY, indices=idx),
            abi.HostColumn(abi.VARCHAR, words, encoding=abi.DICTIONARY, indices=rng.integers(0, 3, n).astype(np.int32)),
            abi.HostColumn(abi.DOUBLE, np.array([2.5]), encoding=abi.CONSTANT),
            abi.HostColumn(abi.INTEGER, np.array([7], dtype=np.int32), valid=[False], encoding=abi.CONSTANT)]
    batch = abi.HostBatch(cols, n)
    offsets = [0, 3000, n]
    assert vx.presto_serialize(batch, offsets) == oracle.presto_serialize(batch, offsets)


def test_presto_page_timestamp_out_of_range_is_a_user_error(oracle, vx):
    ts = np.array([[2 ** 62, 0]], dtype=np.int64)
    batch = abi.HostBatch([abi.HostColumn(abi.TIMESTAMP, ts)])
    for impl in (oracle, vx):
        with pytest.raises(Exception) as e:
            impl.presto_serialize(batch, [0, 1])
        assert "milliseconds" in str(e.value)
    assert vx.presto_serialize(batch, [0, 1], flags=abi.PAGE_LOSSLESS_TIMESTAMP) == \
        oracle.presto_serialize(batch, [0, 1], flags=abi.PAGE_LOSSLESS_TIMESTAMP)


def _expect_rows(py, rows, lossless):
    """Columns of presto_page_reader.random_page_batch in the order 'rows', TIMESTAMP as the
    reader of the wire format sees it: {seconds, nanos} after Timestamp::fromMillis(toMillis)
    unless lossless."""
    out = []
    for kind_values, valid in py:
        out.append(([kind_values[r] for r in rows], [bool(valid[r]) for r in rows]))
    if not lossless:
        vals, valid = out[-1]
        conv = []
        for s, ns in vals:
            ms = s * 1000 + ns // 1000000
            conv.append((ms // 1000, (ms % 1000) * 1000000))  # floor semantics == fromMillis
        out[-1] = (conv, valid)
    return out


@pytest.mark.parametrize("flags", [0, abi.PAGE_CHECKSUM, abi.PAGE_LOSSLESS_TIMESTAMP])
def test_presto_pages_deserialize_to_device_columns(oracle, vx, flags):
    """vx355_presto_deserialize (PrestoVectorSerde::deserialize): pages written by the oracle's
    writer and by the GPU's come back as the rows they were made of - every kind, null bitmaps
    present and absent, several pages appended (sizes around 8 / 64 rows and the empty page),
    strings on both sides of the inline limit, checksums verified."""
    from presto_page_reader import random_page_batch
    rng = np.random.default_rng(4040)
    n = 7000
    batch, py = random_page_batch(rng, n)
    kinds = [c.kind for c in batch.columns]
    rows = rng.permutation(n).astype(np.int32)
    offsets = [0, 0, 1, 9, 73, 137, 2185, 2185, n]
    lossless = bool(flags & abi.PAGE_LOSSLESS_TIMESTAMP)
    want = _expect_rows(py, list(rows), lossless)
    for writer in (oracle, vx):
        pages = writer.presto_serialize(batch, offsets, rows, flags)
        got_n, got = vx.presto_deserialize(pages, kinds, flags & abi.PAGE_LOSSLESS_TIMESTAMP)
        assert got_n == n
        for c, kind in enumerate(kinds):
            gv, gvalid = got[c]
            wv, wvalid = want[c]
            assert list(gvalid) == wvalid, (writer.__name__, c)
            for r in range(n):
                if not wvalid[r]:
                    continue
                g = tuple(int(x) for x in gv[r]) if kind == abi.TIMESTAMP else gv[r]
                assert g == wv[r], (writer.__name__, c, r)
    # a batch without nulls: no bitmaps on the wire
    dense, dpy = random_page_batch(rng, 3000, with_nulls=False)
    pages = oracle.presto_serialize(dense, [0, 3000], flags=flags)
    got_n, got = vx.presto_deserialize(pages, kinds, flags & abi.PAGE_LOSSLESS_TIMESTAMP)
    assert got_n == 3000 and all(all(valid) for _, valid in got)
    assert [int(x) for x in got[0][0]] == dpy[0][0]


@pytest.mark.parametrize("name", ["LZ4", "SNAPPY", "ZSTD", "ZLIB", "GZIP"])
def test_compressed_presto_pages_deserialize_to_device_columns(oracle, vx, name):
    """PrestoVectorSerde::deserialize with PrestoOptions::compressionKind set
    (serializers/PrestoSerializer.cpp:144-145,185-199): pages a compressing exchange sent - bodies
    compressed by the library's own writer step and by an independent codec (pyarrow / zlib), with and
    without checksums, mixed with pages that travelled uncompressed - come back as the rows they were made of."""
    from presto_page_reader import random_page_batch
    from test_page_compression import KINDS, assemble, independent_compress
    kind = KINDS[name]
    rng = np.random.default_rng(6060)
    n = 6000
    batch, py = random_page_batch(rng, n)
    kinds = [c.kind for c in batch.columns]
    offsets = [0, 1, 700, 701, 4000, n]
    want = _expect_rows(py, list(range(n)), False)
    for flags in (0, abi.PAGE_CHECKSUM):
        plain = oracle.presto_serialize(batch, offsets, None, flags)
        ours = [vx.presto_compress_page(p, kind) for p in plain]
        theirs = [assemble(p, name, independent_compress(name, p[21:])) if len(p) > 200 else p for p in plain]
        mixed = [ours[0], plain[1], theirs[2], ours[3], plain[4]]
        assert any(p[4] & 1 for p in ours) and any(p[4] & 1 for p in theirs)
        for pages in (ours, theirs, mixed):
            got_n, got = vx.presto_deserialize(pages, kinds, abi.page_compression(kind))
            assert got_n == n
            for c, k in enumerate(kinds):
                gv, gvalid = got[c]
                wv, wvalid = want[c]
                assert list(gvalid) == wvalid, (name, c)
                for r in range(n):
                    if wvalid[r]:
                        g = tuple(int(x) for x in gv[r]) if k == abi.TIMESTAMP else gv[r]
                        assert g == wv[r], (name, c, r)
    # the reader's configuration names the codec; a corrupt body is the sender's fault
    with pytest.raises(vx.Vx355Error) as e:
        vx.presto_deserialize(ours, kinds, 0)
    assert e.value.status == abi.EINVAL
    big = bytearray(next(p for p in ours if p[4] & 1 and len(p) > 500))
    big[300] ^= 0x10
    with pytest.raises(vx.Vx355Error) as e:
        vx.presto_deserialize([bytes(big)], kinds, abi.page_compression(kind))
    assert e.value.status == abi.EUSER


def test_presto_deserialize_rejects_corrupt_and_mismatched_pages(oracle, vx):
    from presto_page_reader import random_page_batch
    rng = np.random.default_rng(5050)
    batch, _ = random_page_batch(rng, 500)
    kinds = [c.kind for c in batch.columns]
    (page,) = oracle.presto_serialize(batch, [0, 500], flags=abi.PAGE_CHECKSUM)
    broken = bytearray(page)
    broken[len(broken) // 2] ^= 0x40
    with pytest.raises(vx.Vx355Error) as e:
        vx.presto_deserialize([bytes(broken)], kinds)
    assert e.value.status == abi.EUSER and "corrupted" in str(e.value)
    with pytest.raises(vx.Vx355Error) as e:
        vx.presto_deserialize([page], [abi.INTEGER] + kinds[1:])      # BIGINT column read as INTEGER
    assert e.value.status == abi.EUSER
    with pytest.raises(vx.Vx355Error) as e:
        vx.presto_deserialize([page[:-3]], kinds)                       # truncated
    assert e.value.status == abi.EUSER
    assert vx.presto_deserialize([], kinds)[0] == 0
    # string offsets that run backwards are refused before any kernel trusts them
    (spage,) = oracle.presto_serialize(abi.HostBatch([abi.HostColumn(abi.VARCHAR, [b"abc", b"de", b"f"])]), [0, 3])
    at = 25 + 4 + 14 + 4     # header, column count | name length, name, row count -> first end offset
    assert spage[at:at + 12] == (3).to_bytes(4, "little") + (5).to_bytes(4, "little") + (6).to_bytes(4, "little")
    evil = spage[:at + 4] + (2).to_bytes(4, "little") + spage[at + 8:]
    with pytest.raises(vx.Vx355Error) as e:
        vx.presto_deserialize([evil], [abi.VARCHAR])
    assert e.value.status == abi.EUSER and "offsets" in str(e.value)


def _column_bytes(oracle, column):
    """The wire bytes of one flat column (header, row count, nulls, values) from the oracle writer."""
    n = len(column.valid) if column.valid is not None else column.num_rows
    (page,) = oracle.presto_serialize(abi.HostBatch([column], n), [0, n])
    return page[25:]


def _frame(columns, n):
    body = len(columns).to_bytes(4, "little") + b"".join(columns)
    return n.to_bytes(4, "little") + b"\x00" + len(body).to_bytes(4, "little") * 2 + bytes(8) + body


def test_presto_deserialize_reads_rle_and_dictionary_columns(oracle, vx):
    """Pages from writers that keep encodings (Presto's Java workers, the batch serializer):
    RLE = a one-row nested column repeated, DICTIONARY = nested dictionary + int32 indices + 24
    bytes of instance id (VectorStream::flush, serializers/VectorStream.cpp:210-232)."""
    rng = np.random.default_rng(6060)
    n, d = 1000, 37
    words = [b"", b"dict", b"a dictionary entry beyond twelve bytes", b"x" * 12, b"y" * 13]
    dict_s = [words[i % len(words)] + str(i).encode() for i in range(d)]
    dict_valid = rng.random(d) > 0.2
    dict_i = rng.integers(-1000, 1000, d).astype(np.int64)
    idx_s = rng.integers(0, d, n).astype(np.int32)
    idx_i = rng.integers(0, d, n).astype(np.int32)
    rle = lambda child: (3).to_bytes(4, "little") + b"RLE" + n.to_bytes(4, "little") + child
    dic = lambda child, idx: (10).to_bytes(4, "little") + b"DICTIONARY" + n.to_bytes(4, "little") + child + idx.tobytes() + bytes(24)
    cols = [
        rle(_column_bytes(oracle, abi.HostColumn(abi.DOUBLE, np.array([2.5])))),
        rle(_column_bytes(oracle, abi.HostColumn(abi.BIGINT, np.array([7], dtype=np.int64), valid=np.array([False])))),
        rle(_column_bytes(oracle, abi.HostColumn(abi.VARCHAR, [b"a constant longer than twelve bytes"]))),
        dic(_column_bytes(oracle, abi.HostColumn(abi.VARCHAR, dict_s, valid=dict_valid)), idx_s),
        dic(_column_bytes(oracle, abi.HostColumn(abi.BIGINT, dict_i)), idx_i),
        _column_bytes(oracle, abi.HostColumn(abi.INTEGER, np.arange(n, dtype=np.int32))),
    ]
    kinds = [abi.DOUBLE, abi.BIGINT, abi.VARCHAR, abi.VARCHAR, abi.BIGINT, abi.INTEGER]
    page = _frame(cols, n)
    got_n, got = vx.presto_deserialize([page, page], kinds)
    assert got_n == 2 * n
    for half in (0, n):
        assert all(got[0][1][half:half + n]) and (np.asarray(got[0][0][half:half + n]) == 2.5).all()
        assert not any(got[1][1][half:half + n])
        assert all(v == b"a constant longer than twelve bytes" for v in got[2][0][half:half + n])
        for r in range(n):
            assert bool(got[3][1][half + r]) == bool(dict_valid[idx_s[r]])
            if dict_valid[idx_s[r]]:
                assert got[3][0][half + r] == dict_s[idx_s[r]]
        assert (np.asarray(got[4][0][half:half + n]) == dict_i[idx_i]).all() and all(got[4][1][half:half + n])
        assert (np.asarray(got[5][0][half:half + n]) == np.arange(n)).all()
    # an index past the dictionary is refused
    bad = _frame([dic(_column_bytes(oracle, abi.HostColumn(abi.BIGINT, dict_i)), np.full(n, d, dtype=np.int32))], n)
    with pytest.raises(vx.Vx355Error) as e:
        vx.presto_deserialize([bad], [abi.BIGINT])
    assert e.value.status == abi.EUSER


def test_compose_indices_is_a_dictionary_over_a_dictionary(vx):
    """vx355_compose_indices: out[i] = inner[outer[i]] (wrapChild over a wrapped vector,
    exec/OperatorUtils.cpp:393-422), host and device memory; an index outside the inner vector is refused."""
    rng = np.random.default_rng(91)
    for n_inner, n in ((1, 1), (1000, 64), (200_000, 1_000_003)):
        inner = rng.integers(0, 1 << 30, n_inner).astype(np.int32)
        outer = rng.integers(0, n_inner, n).astype(np.int32)
        assert (vx.compose_indices(inner, outer) == inner[outer]).all()
        di, do = vx.DeviceArray(inner), vx.DeviceArray(outer)
        out = vx.DeviceArray(np.zeros(n, dtype=np.int32))
        vx.compose_indices_device(di.ptr, n_inner, do.ptr, n, out.ptr)
        assert (out.to_host() == inner[outer]).all()
    assert len(vx.compose_indices(np.zeros(3, np.int32), np.zeros(0, np.int32))) == 0
    with pytest.raises(vx.Vx355Error) as e:
        vx.compose_indices(np.arange(10, dtype=np.int32), np.array([3, 10], dtype=np.int32))
    assert e.value.status == abi.EINVAL


def test_hbm_ceiling_kernels_report_plausible_rates(vx):
    """The library's own read-only-stream and copy kernels (bench.py quotes them next to the 8 TB/s
    datasheet peak): both land between 1 and 8 TB/s on an MI355X, the read-only stream not below
    the copy."""
    read = vx.hbm_ceiling(abi.CEILING_READ, 2 << 30, 3)
    copy = vx.hbm_ceiling(abi.CEILING_COPY, 1 << 30, 3)
    assert 1000 < copy < 8000 and 1000 < read < 8000, (read, copy)
    # the seven column streams of TPC-H Q1's scan, nothing computed: what k_agg_fast competes with
    columns = vx.hbm_ceiling(abi.CEILING_READ_COLUMNS, 2 << 30, 3)
    assert 1000 < columns < 8000, columns
    print("hbm ceilings GB/s: read", round(read), "copy", round(copy), "seven column streams", round(columns))
    with pytest.raises(vx.Vx355Error) as e:
        vx.hbm_ceiling(3, 1 << 30, 1)
    assert e.value.status == abi.EINVAL


@pytest.mark.parametrize("device_resident", [False, True])
def test_presto_pages_with_struct_columns_equal_the_oracle_and_read_back(oracle, vx, device_resident):
    """ROW columns (VX355_ROW; the intermediate type of avg is ROW(DOUBLE, BIGINT),
    AverageAggregateBase.h:66-260) through vx355_presto_serialize and vx355_presto_deserialize:
    byte for byte the oracle's pages (VectorStream::flush ROW branch + serializeRowVector), decoded
    by the independent reader, and read back into HBM columns - null structs, nulls inside fields, a
    string field, a struct whose every row is null, a row list, page sizes around the null byte and
    the 2048-row tile, empty pages, checksums."""
    from presto_page_reader import check_pages_decode_to_rows, random_row_page_batch
    rng = np.random.default_rng(909)
    n = 7000
    batch, py, kinds = random_row_page_batch(rng, n)
    src = batch
    if device_resident:
        cols = []
        for c in batch.columns:
            if isinstance(c, abi.HostRowColumn):
                cols.append(abi.HostRowColumn([vx.DeviceColumn(k) for k in c.children], c.valid))
            else:
                cols.append(vx.DeviceColumn(c))
        src = abi.HostBatch(cols, n)
    rows = rng.permutation(n).astype(np.int32)
    offsets = [0, 0, 1, 8, 17, 2065, 4113, 4113, 4114 + 2047, n]
    for flags in (0, abi.PAGE_CHECKSUM):
        exp = oracle.presto_serialize(batch, offsets, rows, flags)
        got = vx.presto_serialize(src, offsets, rows, flags)
        assert [len(g) for g in got] == [len(e) for e in exp]
        assert got == exp, f"flags {flags}"
    pages = vx.presto_serialize(src, offsets, rows, abi.PAGE_CHECKSUM)
    check_pages_decode_to_rows(pages, py, kinds, offsets, rows)
    assert vx.presto_serialize(src, [0, 3000, n]) == oracle.presto_serialize(batch, [0, 3000, n])
    # ... and back: the reader's columns hold a row per struct row, fields null where the struct is
    for writer_pages in (pages, oracle.presto_serialize(batch, offsets, rows)):
        got_n, got = vx.presto_deserialize(writer_pages, kinds)
        assert got_n == n
        for c, kind in enumerate(kinds):
            src_vals, src_valid = py[c]
            gv, gvalid = got[c]
            want_valid = [bool(src_valid[r]) for r in rows]
            assert list(gvalid) == want_valid, c
            if not isinstance(kind, tuple):
                assert [int(x) for x in gv] == [src_vals[r] for r in rows]
                continue
            for f, (fv, fvalid) in enumerate(gv):
                sv, svalid = src_vals[f]
                for i, r in enumerate(rows):
                    assert bool(fvalid[i]) == (want_valid[i] and bool(svalid[r])), (c, f, i)
                    if fvalid[i]:
                        assert fv[i] == sv[r], (c, f, i)
    # a ROW inside a ROW is refused
    nested = abi.HostBatch([abi.HostRowColumn([abi.HostRowColumn([abi.HostColumn(abi.BIGINT, np.arange(4))])])])
    with pytest.raises(vx.Vx355Error) as e:
        vx.presto_serialize(nested, [0, 4])
    assert e.value.status == abi.EUNSUPPORTED
