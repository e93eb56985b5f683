"""An independent reader of the PrestoPage wire format, written from the format description
(prestodb.io/docs/current/develop/serialized-page.html; the reader side of the reference is
serializers/PrestoSerializerDeserializationUtils.cpp): the parity tests decode what the GPU
and the oracle wrote and compare with the input rows."""
import struct
import zlib

import numpy as np

from velox_amd import abi

_FIXED = {"BYTE_ARRAY": 1, "SHORT_ARRAY": 2, "INT_ARRAY": 4, "LONG_ARRAY": 8}


def is_row(kind):
    """A struct column is described as ("row", [child kinds])."""
    return isinstance(kind, tuple) and kind[0] == "row"


def _read_column(page, pos, kind, num_rows, lossless_timestamp):
    """One column stream at 'pos' -> (values, valid, position behind it). A ROW column's values are
    the children's (values, valid) lists, expanded to one entry per struct row (None / False where
    the struct is null)."""
    (name_len,) = struct.unpack_from("<i", page, pos)
    name = page[pos + 4:pos + 4 + name_len].decode()
    pos += 4 + name_len
    if is_row(kind):
        assert name == "ROW"
        (num_children,) = struct.unpack_from("<i", page, pos)
        pos += 4
        assert num_children == len(kind[1])
        # the children hold one row per NON-NULL struct: their row count is only known at the end
        child_at = pos
        children = []
        for child_kind in kind[1]:
            (clen,) = struct.unpack_from("<i", page, pos)
            (child_rows,) = struct.unpack_from("<i", page, pos + 4 + clen)
            cv, cvalid, pos = _read_column(page, pos, child_kind, child_rows, lossless_timestamp)
            children.append((cv, cvalid, child_rows))
        del child_at
        (n,) = struct.unpack_from("<i", page, pos)
        pos += 4
        assert n == num_rows
        offsets = struct.unpack_from(f"<{n + 1}i", page, pos)
        pos += 4 * (n + 1)
        assert offsets[0] == 0
        has_nulls = page[pos]
        pos += 1
        valid = [True] * n
        if has_nulls:
            for r in range(n):
                valid[r] = not ((page[pos + r // 8] >> (7 - r % 8)) & 1)
            if n % 8:
                assert page[pos + (n - 1) // 8] & ((1 << (8 - n % 8)) - 1) == 0
            pos += (n + 7) // 8
            assert not all(valid), "hasNulls without a null"
        non_null = sum(valid)
        for r in range(n):
            assert offsets[r + 1] - offsets[r] == (1 if valid[r] else 0), "ROW offsets count the non-null structs"
        expanded = []
        for cv, cvalid, child_rows in children:
            assert child_rows == non_null, "a child holds the rows of the non-null structs"
            ev, evalid, at = [None] * n, [False] * n, 0
            for r in range(n):
                if valid[r]:
                    ev[r], evalid[r] = cv[at], cvalid[at]
                    at += 1
            expanded.append((ev, evalid))
        return expanded, valid, pos
    (n,) = struct.unpack_from("<i", page, pos)
    pos += 4
    assert n == num_rows
    ends = None
    if name == "VARIABLE_WIDTH":
        ends = struct.unpack_from(f"<{n}i", page, pos)
        pos += 4 * n
    has_nulls = page[pos]
    pos += 1
    valid = [True] * n
    if has_nulls:
        for r in range(n):
            valid[r] = not ((page[pos + r // 8] >> (7 - r % 8)) & 1)
        # padding bits of the last byte are zero
        if n % 8:
            assert page[pos + (n - 1) // 8] & ((1 << (8 - n % 8)) - 1) == 0
        pos += (n + 7) // 8
        assert not all(valid), "hasNulls without a null"
    values = [None] * n
    if name == "VARIABLE_WIDTH":
        (total,) = struct.unpack_from("<i", page, pos)
        pos += 4
        assert kind in (abi.VARCHAR, abi.VARBINARY)
        assert total == (ends[-1] if n else 0)
        prev = 0
        for r in range(n):
            if valid[r]:
                values[r] = page[pos + prev:pos + ends[r]]
            else:
                assert ends[r] == prev
            prev = ends[r]
        pos += total
    else:
        w = _FIXED[name]
        if kind == abi.TIMESTAMP and lossless_timestamp:
            w = 16
        for r in range(n):
            if not valid[r]:
                continue
            raw = page[pos:pos + w]
            pos += w
            if kind == abi.BOOLEAN:
                assert name == "BYTE_ARRAY" and raw[0] in (0, 1)
                values[r] = bool(raw[0])
            elif kind == abi.REAL:
                assert name == "INT_ARRAY"
                values[r] = np.frombuffer(raw, dtype=np.float32)[0]
            elif kind == abi.DOUBLE:
                assert name == "LONG_ARRAY"
                values[r] = np.frombuffer(raw, dtype=np.float64)[0]
            elif kind == abi.TIMESTAMP:
                values[r] = struct.unpack("<qQ", raw) if lossless_timestamp else struct.unpack("<q", raw)[0]
            else:
                values[r] = int.from_bytes(raw, "little", signed=True)
    return values, valid, pos


def read_page(page, kinds, lossless_timestamp=False):
    """-> (num_rows, [(values list, valid list)] per column). kinds: vx355 type kinds; a struct
    column is ("row", [child kinds]) and comes back as ([(values, valid) per child], valid)."""
    num_rows, codec, uncompressed, size, checksum = struct.unpack_from("<ibiiq", page, 0)
    assert uncompressed == size == len(page) - 21
    if codec & 4:
        crc = zlib.crc32(page[21:] + page[4:5] + page[0:4] + page[5:9]) & 0xffffffff
        assert checksum == crc, "page checksum"
    else:
        assert codec == 0 and checksum == 0
    pos = 21
    (num_cols,) = struct.unpack_from("<i", page, pos)
    pos += 4
    assert num_cols == len(kinds)
    cols = []
    for kind in kinds:
        values, valid, pos = _read_column(page, pos, kind, num_rows, lossless_timestamp)
        cols.append((values, valid))
    assert pos == len(page)
    return num_rows, cols


def random_page_batch(rng, n, with_nulls=True):
    """-> (HostBatch of every serializable kind, [(python values, valid)] per column; the
    TIMESTAMP column's python values are (seconds, nanos) pairs)."""
    words = [b"", b"x", b"twelve bytes", b"thirteen byte", b"a considerably longer string value " * 3]

    def valid():
        return (rng.random(n) > 0.25) if with_nulls else None

    ts = np.stack([rng.integers(-10**9, 10**9, n), rng.integers(0, 10**9, n)], axis=1).astype(np.int64)
    data = [(abi.BIGINT, rng.integers(-2**62, 2**62, n).astype(np.int64)),
            (abi.INTEGER, rng.integers(-2**31, 2**31, n).astype(np.int32)),
            (abi.SMALLINT, rng.integers(-2**15, 2**15, n).astype(np.int16)),
            (abi.TINYINT, rng.integers(-128, 128, n).astype(np.int8)),
            (abi.BOOLEAN, rng.random(n) > 0.5),
            (abi.REAL, rng.random(n).astype(np.float32)),
            (abi.DOUBLE, rng.random(n)),
            (abi.VARCHAR, [words[i] for i in rng.integers(0, len(words), n)]),
            (abi.TIMESTAMP, ts)]
    cols, py = [], []
    for j, (kind, values) in enumerate(data):
        v = None if j == 6 else valid()  # the DOUBLE column never has nulls: no bitmap on the wire
        cols.append(abi.HostColumn(kind, values, valid=v))
        if kind == abi.TIMESTAMP:
            pv = [(int(s), int(ns)) for s, ns in values]
        elif kind in (abi.VARCHAR, abi.REAL, abi.DOUBLE):
            pv = list(values)
        elif kind == abi.BOOLEAN:
            pv = [bool(x) for x in values]
        else:
            pv = [int(x) for x in values]
        py.append((pv, np.ones(n, bool) if v is None else np.asarray(v, bool)))
    return abi.HostBatch(cols), py


def random_row_page_batch(rng, n):
    """-> (HostBatch with struct columns, python values, kinds): a key column, avg's intermediate
    ROW(DOUBLE, BIGINT) with null structs and nulls inside, a struct with a string field and without
    null structs, a struct whose every row is null."""
    words = [b"", b"x", b"twelve bytes", b"thirteen byte", b"a considerably longer string value " * 2]
    key = rng.integers(-2**40, 2**40, n).astype(np.int64)
    sums = rng.random(n)
    counts = rng.integers(0, 1000, n).astype(np.int64)
    avg_valid = rng.random(n) > 0.3
    count_valid = rng.random(n) > 0.1
    strs = [words[i] for i in rng.integers(0, len(words), n)]
    str_valid = rng.random(n) > 0.2
    small = rng.integers(-2**15, 2**15, n).astype(np.int16)
    cols = [abi.HostColumn(abi.BIGINT, key),
            abi.HostRowColumn([abi.HostColumn(abi.DOUBLE, sums), abi.HostColumn(abi.BIGINT, counts, valid=count_valid)],
                              valid=avg_valid),
            abi.HostRowColumn([abi.HostColumn(abi.VARCHAR, strs, valid=str_valid), abi.HostColumn(abi.SMALLINT, small)]),
            abi.HostRowColumn([abi.HostColumn(abi.BIGINT, counts)], valid=np.zeros(n, bool))]
    kinds = [abi.BIGINT, ("row", [abi.DOUBLE, abi.BIGINT]), ("row", [abi.VARCHAR, abi.SMALLINT]), ("row", [abi.BIGINT])]
    ones = np.ones(n, bool)
    py = [([int(x) for x in key], ones),
          ([(list(sums), ones), ([int(x) for x in counts], count_valid)], avg_valid),
          ([(strs, str_valid), ([int(x) for x in small], ones)], ones),
          ([([int(x) for x in counts], ones)], np.zeros(n, bool))]
    return abi.HostBatch(cols), py, kinds


def millis(py):
    """The TIMESTAMP column (last) as Timestamp::toMillis values."""
    vals, valid = py[-1]
    return py[:-1] + [([s * 1000 + ns // 1000000 for s, ns in vals], valid)]


def check_pages_decode_to_rows(pages, batch_cols_py, kinds, offsets, rows, lossless=False):
    for p, page in enumerate(pages):
        lo, hi = offsets[p], offsets[p + 1]
        if lo == hi:
            assert page == b""
            continue
        n, cols = read_page(page, kinds, lossless)
        assert n == hi - lo
        for c, (values, valid) in enumerate(cols):
            src_vals, src_valid = batch_cols_py[c]
            for i in range(n):
                r = rows[lo + i] if rows is not None else lo + i
                assert valid[i] == bool(src_valid[r]), (p, c, i)
                if not valid[i]:
                    continue
                if is_row(kinds[c]):
                    # a struct: every field against the source field at the same row
                    for f, (fv, fvalid) in enumerate(values):
                        sv, svalid = src_vals[f]
                        assert fvalid[i] == bool(svalid[r]), (p, c, f, i)
                        if fvalid[i]:
                            assert fv[i] == sv[r], (p, c, f, i)
                else:
                    assert values[i] == src_vals[r], (p, c, i)
