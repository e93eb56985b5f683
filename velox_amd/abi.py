"""ctypes mirror of include/vx355.h (structs, enums) and numpy helpers that lay
out Velox-shaped buffers: bit-packed null / boolean bitmaps
(common/base/Nulls.h:26-38, 1 = not null) and 16-byte StringViews
(type/StringView.h:76-77).

Pure data-layout code: no compute happens here.
"""
import ctypes as C

import numpy as np

# vx355_status
OK, EUSER, EUNSUPPORTED, ENOMEM, EINTERNAL, EINVAL = range(6)

# vx355_type_kind == velox::TypeKind (type/TypeKind.h:41-52)
BOOLEAN, TINYINT, SMALLINT, INTEGER, BIGINT, REAL, DOUBLE, VARCHAR, VARBINARY, TIMESTAMP = range(10)
ROW = 32   # PrestoPage entry points only: a struct column (HostRowColumn)

# vx355_encoding
FLAT, CONSTANT, DICTIONARY = range(3)
# vx355_mem
MEM_HOST, MEM_DEVICE = 0, 1

# vx355_agg_kind
AGG_SUM, AGG_COUNT, AGG_COUNT_STAR, AGG_MIN, AGG_MAX, AGG_AVG = range(6)
# vx355_agg_step == core::AggregationNode::Step (core/PlanNode.h:1122-1131)
STEP_PARTIAL, STEP_FINAL, STEP_INTERMEDIATE, STEP_SINGLE = range(4)

# vx355_join_type == core::JoinType (core/PlanNode.h:3081-3165)
(JOIN_INNER, JOIN_LEFT, JOIN_RIGHT, JOIN_FULL, JOIN_LEFT_SEMI_FILTER,
 JOIN_COUNTING_LEFT_SEMI_FILTER, JOIN_LEFT_SEMI_PROJECT, JOIN_RIGHT_SEMI_FILTER,
 JOIN_RIGHT_SEMI_PROJECT, JOIN_ANTI, JOIN_COUNTING_ANTI, JOIN_RIGHT_ANTI) = range(12)

# vx355_partition_kind
PART_MODULO, PART_BIT_RANGE, PART_LOCAL_MODULO, PART_LOCAL_BIT_RANGE = range(4)

# BaseHashTable::HashMode as reported in stats
MODE_HASH, MODE_ARRAY, MODE_NORMALIZED_KEY = 0, 1, 2

KIND_DTYPE = {
    TINYINT: np.int8, SMALLINT: np.int16, INTEGER: np.int32, BIGINT: np.int64,
    REAL: np.float32, DOUBLE: np.float64,
}
KIND_WIDTH = {BOOLEAN: 0, TINYINT: 1, SMALLINT: 2, INTEGER: 4, BIGINT: 8, REAL: 4, DOUBLE: 8,
              VARCHAR: 16, VARBINARY: 16, TIMESTAMP: 16}


class Column(C.Structure):
    _fields_ = [("type_kind", C.c_int32), ("encoding", C.c_int32), ("values", C.c_void_p),
                ("nulls", C.c_void_p), ("indices", C.c_void_p), ("base_size", C.c_int32),
                ("mem", C.c_int32)]


class Batch(C.Structure):
    _fields_ = [("num_rows", C.c_int32), ("num_cols", C.c_int32), ("cols", C.POINTER(Column))]


class OutColumn(C.Structure):
    _fields_ = [("type_kind", C.c_int32), ("mem", C.c_int32), ("values", C.c_void_p),
                ("nulls", C.c_void_p)]


class ValueIdSpec(C.Structure):
    _fields_ = [("min", C.c_int64), ("max", C.c_int64), ("multiplier", C.c_uint64)]


class AggFn(C.Structure):
    _fields_ = [("kind", C.c_int32), ("input_col", C.c_int32), ("input_col2", C.c_int32),
                ("input_type", C.c_int32), ("mask_col", C.c_int32), ("flags", C.c_int32)]


class AggSpec(C.Structure):
    _fields_ = [("num_keys", C.c_int32), ("key_cols", C.POINTER(C.c_int32)),
                ("key_types", C.POINTER(C.c_int32)), ("num_aggs", C.c_int32),
                ("aggs", C.POINTER(AggFn)), ("step", C.c_int32), ("ignore_null_keys", C.c_int32),
                ("flags", C.c_int32), ("pad", C.c_int32)]


AGG_UNORDERED_OUTPUT = 1
AGG_FN_DISTINCT = 1  # vx355_agg_fn.flags
PAGE_CHECKSUM = 1
PAGE_LOSSLESS_TIMESTAMP = 2
# vx355_compression_kind = common::CompressionKind (common/compression/Compression.h:28-37)
COMPRESSION_NONE, COMPRESSION_ZLIB, COMPRESSION_SNAPPY, COMPRESSION_LZO, COMPRESSION_ZSTD, COMPRESSION_LZ4, COMPRESSION_GZIP = range(7)


def page_compression(kind):
    """VX355_PAGE_COMPRESSION(kind): the exchange's compression kind inside a page-flags word."""
    return (kind & 0xff) << 8


class AggStats(C.Structure):
    _fields_ = [("num_groups", C.c_int64), ("capacity", C.c_int64), ("num_rehashes", C.c_int64),
                ("hash_mode", C.c_int32), ("reserved", C.c_int32), ("input_rows", C.c_int64),
                ("deferred_rows", C.c_int64), ("radix_launches", C.c_int64), ("table_bytes", C.c_int64),
                ("num_flushes", C.c_int64), ("compact_record_launches", C.c_int64)]


class KeyFilter(C.Structure):
    _fields_ = [("kind", C.c_int32), ("pad", C.c_int32), ("min", C.c_int64), ("max", C.c_int64),
                ("num_distinct", C.c_int64)]


KEY_FILTER_NONE, KEY_FILTER_VALUES, KEY_FILTER_BLOOM = 0, 1, 2
CEILING_READ, CEILING_COPY, CEILING_READ_COLUMNS = 0, 1, 2


class JoinBuildSpec(C.Structure):
    _fields_ = [("num_keys", C.c_int32), ("key_cols", C.POINTER(C.c_int32)),
                ("key_types", C.POINTER(C.c_int32)), ("num_dependents", C.c_int32),
                ("dependent_cols", C.POINTER(C.c_int32)),
                ("dependent_types", C.POINTER(C.c_int32)), ("join_type", C.c_int32),
                ("null_aware", C.c_int32), ("null_as_value", C.c_int32), ("drop_duplicates", C.c_int32)]


class JoinTableStats(C.Structure):
    _fields_ = [("num_rows", C.c_int64), ("num_distinct", C.c_int64), ("capacity", C.c_int64),
                ("hash_mode", C.c_int32), ("has_duplicates", C.c_int32)]


class JoinProbeSpec(C.Structure):
    _fields_ = [("num_keys", C.c_int32), ("key_cols", C.POINTER(C.c_int32)),
                ("join_type", C.c_int32), ("null_aware", C.c_int32), ("null_as_value", C.c_int32),
                ("pad", C.c_int32)]


# vx355_cmp
CMP_EQ, CMP_NE, CMP_LT, CMP_LE, CMP_GT, CMP_GE = range(6)


class FilterTerm(C.Structure):
    _fields_ = [("col", C.c_int32), ("cmp", C.c_int32), ("const_kind", C.c_int32),
                ("str_size", C.c_int32), ("i64", C.c_int64), ("f64", C.c_double),
                ("str", C.c_char * 16)]


class JoinFilterTerm(C.Structure):
    _fields_ = [("left_side", C.c_int32), ("left_col", C.c_int32), ("cmp", C.c_int32),
                ("right_kind", C.c_int32), ("right_col", C.c_int32), ("const_kind", C.c_int32),
                ("str_size", C.c_int32), ("pad", C.c_int32), ("i64", C.c_int64), ("f64", C.c_double),
                ("str", C.c_char * 16)]


BUILD_COL_MATCH = -1   # VX355_BUILD_COL_MATCH: the 'match' column of RIGHT_SEMI_PROJECT


def join_filter_terms(terms):
    """[(left, cmp, right)] -> ctypes array. left / right: ("probe", col), ("build", dep index) or a
    constant (int -> BIGINT, float -> DOUBLE, bytes -> VARCHAR); left must be a column."""
    arr = (JoinFilterTerm * max(1, len(terms)))()
    for i, (left, cmp_, right) in enumerate(terms):
        t = arr[i]
        t.left_side = 0 if left[0] == "probe" else 1
        t.left_col = left[1]
        t.cmp = cmp_
        if isinstance(right, tuple):
            t.right_kind = 1 if right[0] == "probe" else 2
            t.right_col = right[1]
        elif isinstance(right, bytes):
            t.right_kind, t.const_kind, t.str_size, t.str = 0, VARCHAR, len(right), right
        elif isinstance(right, float):
            t.right_kind, t.const_kind, t.f64 = 0, DOUBLE, right
        else:
            t.right_kind, t.const_kind, t.i64 = 0, BIGINT, int(right)
    return arr


class Factor(C.Structure):
    _fields_ = [("col", C.c_int32), ("pad", C.c_int32), ("scale", C.c_double),
                ("offset", C.c_double)]


class Projection(C.Structure):
    _fields_ = [("num_factors", C.c_int32), ("pad", C.c_int32), ("factors", Factor * 4)]


def filter_terms(terms):
    """[(col, cmp, constant)] -> ctypes array; the constant's python type picks
    const_kind (int -> BIGINT, float -> DOUBLE, bytes -> VARCHAR)."""
    arr = (FilterTerm * max(1, len(terms)))()
    for i, (col, cmp_, const) in enumerate(terms):
        arr[i].col, arr[i].cmp = col, cmp_
        if isinstance(const, bytes):
            arr[i].const_kind, arr[i].str_size, arr[i].str = VARCHAR, len(const), const
        elif isinstance(const, float):
            arr[i].const_kind, arr[i].f64 = DOUBLE, const
        else:
            arr[i].const_kind, arr[i].i64 = BIGINT, int(const)
    return arr


def projections(projs):
    """[[(col, scale, offset), ...], ...] -> ctypes array; col = -1 for a constant factor."""
    arr = (Projection * max(1, len(projs)))()
    for j, factors in enumerate(projs):
        arr[j].num_factors = len(factors)
        for f, (col, scale, offset) in enumerate(factors):
            arr[j].factors[f] = Factor(col, 0, scale, offset)
    return arr


def i32_array(values):
    arr = (C.c_int32 * max(1, len(values)))(*values)
    return arr


def num_words(n):
    return (n + 63) // 64


def pack_bits(mask):
    """bool array -> little-endian uint64 bitmap (bit i of word i/64 = mask[i])."""
    mask = np.asarray(mask, dtype=bool)
    n = len(mask)
    words = num_words(n)
    out = np.zeros(max(1, words) * 8, dtype=np.uint8)
    packed = np.packbits(mask, bitorder="little")
    out[: len(packed)] = packed
    return out.view(np.uint64)


def unpack_bits(words, n):
    bits = np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")
    return bits[:n].astype(bool)


def string_views(strings):
    """list of bytes -> (uint8[n,16] StringViews, keepalive) following
    type/StringView.h: size, 4-byte prefix, 8 inline bytes or a pointer."""
    n = len(strings)
    out = np.zeros((max(n, 1), 16), dtype=np.uint8)
    keep = []
    for i, s in enumerate(strings):
        if s is None:
            continue
        s = bytes(s)
        out[i, 0:4] = np.frombuffer(np.uint32(len(s)).tobytes(), dtype=np.uint8)
        if len(s) <= 12:
            out[i, 4:4 + len(s)] = np.frombuffer(s, dtype=np.uint8)
        else:
            out[i, 4:8] = np.frombuffer(s[:4], dtype=np.uint8)
            buf = C.create_string_buffer(s, len(s))
            keep.append(buf)
            out[i, 8:16] = np.frombuffer(np.uint64(C.addressof(buf)).tobytes(), dtype=np.uint8)
    return out[:n] if n else out[:0], keep


def view_to_bytes(view16):
    """One 16-byte StringView (uint8[16]) -> bytes; inline strings only unless
    the pointer is a live host pointer."""
    size = int(np.frombuffer(view16[0:4].tobytes(), dtype=np.uint32)[0])
    if size <= 12:
        return view16[4:4 + size].tobytes()
    ptr = int(np.frombuffer(view16[8:16].tobytes(), dtype=np.uint64)[0])
    return C.string_at(ptr, size)


class HostColumn:
    """A decoded vector on the host: numpy buffers + the vx355_column describing
    them. values: numpy array (bool for BOOLEAN, list of bytes for VARCHAR,
    (n,2) int64/uint64 pairs for TIMESTAMP). valid: bool array or None."""

    def __init__(self, kind, values, valid=None, encoding=FLAT, indices=None):
        self.kind = kind
        self.encoding = encoding
        self.keep = []
        if kind == BOOLEAN:
            self.values = pack_bits(np.asarray(values, dtype=bool))
            self.base_size = len(values)
        elif kind in (VARCHAR, VARBINARY):
            self.values, keep = string_views(list(values))
            self.values = np.ascontiguousarray(self.values)
            self.keep.append(keep)
            self.base_size = len(values)
        elif kind == TIMESTAMP:
            self.values = np.ascontiguousarray(np.asarray(values, dtype=np.int64).reshape(-1, 2))
            self.base_size = len(self.values)
        else:
            self.values = np.ascontiguousarray(np.asarray(values, dtype=KIND_DTYPE[kind]))
            self.base_size = len(self.values)
        self.indices = None
        if encoding == DICTIONARY:
            self.indices = np.ascontiguousarray(np.asarray(indices, dtype=np.int32))
            self.num_rows = len(self.indices)
        elif encoding == CONSTANT:
            self.num_rows = None
        else:
            self.num_rows = self.base_size
        self.valid = None if valid is None else np.asarray(valid, dtype=bool)
        self.nulls = None if valid is None else pack_bits(self.valid)

    def descriptor(self):
        c = Column()
        c.type_kind = self.kind
        c.encoding = self.encoding
        c.values = self.values.ctypes.data if self.values.size else None
        c.nulls = self.nulls.ctypes.data if self.nulls is not None else None
        c.indices = self.indices.ctypes.data if self.indices is not None else None
        c.base_size = self.base_size if self.encoding == DICTIONARY else 0
        c.mem = MEM_HOST
        return c


class HostRowColumn:
    """A struct column (VX355_ROW) for the PrestoPage entry points: children = HostColumns (or
    device columns with a descriptor()), valid = the struct's own validity (bool array) or None."""

    def __init__(self, children, valid=None):
        self.kind = ROW
        self.encoding = FLAT
        self.children = list(children)
        self.num_rows = next((c.num_rows for c in self.children if c.num_rows is not None), 0)
        self.valid = None if valid is None else np.asarray(valid, dtype=bool)
        self.nulls = None if valid is None else pack_bits(self.valid)
        self._descs = (Column * max(1, len(self.children)))(*[c.descriptor() for c in self.children])

    def descriptor(self):
        c = Column()
        c.type_kind = ROW
        c.encoding = FLAT
        c.values = C.cast(self._descs, C.c_void_p).value
        c.nulls = self.nulls.ctypes.data if self.nulls is not None else None
        c.indices = None
        c.base_size = len(self.children)
        c.mem = MEM_HOST
        return c


class HostBatch:
    """RowVector reduced to decoded children (vx355_batch)."""

    def __init__(self, columns, num_rows=None):
        self.columns = list(columns)
        if num_rows is None:
            num_rows = next((c.num_rows for c in self.columns if c.num_rows is not None), 0)
        self.num_rows = int(num_rows)
        self._descs = (Column * max(1, len(self.columns)))(*[c.descriptor() for c in self.columns])
        self.batch = Batch(self.num_rows, len(self.columns), self._descs)

    def ref(self):
        return C.byref(self.batch)


class OutBuffers:
    """Caller-allocated flat output columns on the host."""

    def __init__(self, kinds, capacity, with_nulls=True):
        self.kinds = list(kinds)
        self.capacity = int(capacity)
        self.values = []
        self.nulls = []
        for k in self.kinds:
            if k == BOOLEAN:
                self.values.append(np.zeros(max(1, num_words(capacity)), dtype=np.uint64))
            else:
                self.values.append(np.zeros(max(1, capacity) * KIND_WIDTH[k], dtype=np.uint8))
            self.nulls.append(np.zeros(max(1, num_words(capacity)), dtype=np.uint64)
                              if with_nulls else None)
        self.descs = (OutColumn * max(1, len(self.kinds)))()
        for i, k in enumerate(self.kinds):
            self.descs[i].type_kind = k
            self.descs[i].mem = MEM_HOST
            self.descs[i].values = self.values[i].ctypes.data
            self.descs[i].nulls = self.nulls[i].ctypes.data if self.nulls[i] is not None else None

    def column(self, i, n):
        """-> (values, valid) as python-friendly arrays for the first n rows."""
        k = self.kinds[i]
        valid = unpack_bits(self.nulls[i], n) if self.nulls[i] is not None else np.ones(n, bool)
        if k == BOOLEAN:
            vals = unpack_bits(self.values[i], n)
        elif k in (VARCHAR, VARBINARY):
            raw = self.values[i][: n * 16].reshape(n, 16)
            sizes = raw[:, 0:4].copy().view(np.uint32).reshape(-1)
            vals = []
            for j in range(n):
                if not valid[j]:
                    vals.append(None)
                elif sizes[j] <= 12:
                    vals.append(view_to_bytes(raw[j]))
                else:
                    # non-inline view of a HOST output column: the pointer leads into a buffer the
                    # operator keeps alive until its next get_output (include/vx355.h, vx355_out_column)
                    ptr = int(raw[j, 8:16].copy().view(np.uint64)[0])
                    vals.append(C.string_at(ptr, int(sizes[j])))
        elif k == TIMESTAMP:
            vals = self.values[i][: n * 16].view(np.int64).reshape(n, 2).copy()
        else:
            vals = self.values[i][: n * KIND_WIDTH[k]].view(KIND_DTYPE[k]).copy()
        return vals, valid
