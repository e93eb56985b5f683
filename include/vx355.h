/*
 * vx355.h — C ABI of libvx355, the MI355X (gfx950) implementation of Velox's
 * HashAggregation / HashBuild / HashProbe hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ types, no
 * torch types, no exceptions. Every entry point names the reference interface
 * it replaces (paths relative to /root/reference/velox). INTEGRATION.md shows
 * the exec::Operator shim a Velox maintainer adds on top of this header.
 *
 * Conventions
 *  - Every function returning int returns a vx355_status. On failure
 *    vx355_last_error() (thread local) describes it. VX355_EUSER maps to
 *    VELOX_USER_FAIL, everything else to VELOX_FAIL; VX355_EUNSUPPORTED at
 *    *_create time means "leave the CPU operator in place"
 *    (cf. experimental/cudf/exec/ToCudf.cpp:230-242).
 *  - Null bitmaps follow common/base/Nulls.h:26-38: bit = 1 means NOT null.
 *  - BOOLEAN values are bit packed (vector/FlatVector.h), VARCHAR values are
 *    16-byte StringView (type/StringView.h:76-77): u32 size, 4-byte prefix,
 *    then 8 more inline bytes (size <= 12) or a pointer (size > 12).
 *  - Inputs are borrowed for the duration of the call. Outputs are written
 *    into caller allocated buffers (host or device, see vx355_mem).
 *  - Handles are single threaded (one Driver thread at a time, exec/Driver.cpp
 *    :538) and each owns one HIP stream; different handles run concurrently
 *    from different threads (no library-wide lock). A vx355_join_table is
 *    immutable after finish and may be shared by any number of probe handles
 *    on any thread.
 */
#ifndef VX355_H_
#define VX355_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VX355_ABI_VERSION 9

typedef enum vx355_status {
  VX355_OK = 0,
  VX355_EUSER = 1,        /* user error, e.g. "integer overflow" in sum(BIGINT) */
  VX355_EUNSUPPORTED = 2, /* type / encoding / join kind not handled on device */
  VX355_ENOMEM = 3,       /* HBM arena exhausted */
  VX355_EINTERNAL = 4,    /* HIP error or broken invariant */
  VX355_EINVAL = 5        /* malformed argument */
} vx355_status;

/* Values equal velox::TypeKind (type/TypeKind.h:41-52). DATE is INTEGER. */
typedef enum vx355_type_kind {
  VX355_BOOLEAN = 0,
  VX355_TINYINT = 1,
  VX355_SMALLINT = 2,
  VX355_INTEGER = 3,
  VX355_BIGINT = 4,
  VX355_REAL = 5,
  VX355_DOUBLE = 6,
  VX355_VARCHAR = 7,
  VX355_VARBINARY = 8,
  VX355_TIMESTAMP = 9,
  /* A struct column (RowVector child of a RowVector), PrestoPage entry points only: the intermediate
   * type of avg is ROW(DOUBLE sum, BIGINT count) (functions/lib/aggregates/AverageAggregateBase.h:
   * 66-260). vx355_column: encoding FLAT, values = the children as a HOST array of base_size
   * vx355_column (scalar kinds, FLAT / CONSTANT / DICTIONARY, in 'mem'), nulls = the struct's own
   * null bitmap. A ROW inside a ROW is VX355_EUNSUPPORTED. */
  VX355_ROW = 32
} vx355_type_kind;

/* vector/VectorEncoding.h: the three encodings DecodedVector reduces to. */
typedef enum vx355_encoding {
  VX355_FLAT = 0,
  VX355_CONSTANT = 1,
  VX355_DICTIONARY = 2
} vx355_encoding;

typedef enum vx355_mem { VX355_MEM_HOST = 0, VX355_MEM_DEVICE = 1 } vx355_mem;

/* One input column, i.e. what DecodedVector (vector/DecodedVector.h) exposes:
 * data(), nulls(&rows), indices(). */
typedef struct vx355_column {
  int32_t type_kind;     /* vx355_type_kind */
  int32_t encoding;      /* vx355_encoding */
  const void* values;    /* FLAT: num_rows values. CONSTANT: 1 value.
                            DICTIONARY: base_size base values. */
  const uint64_t* nulls; /* bit per ROW (top level, after decoding), 1 = valid;
                            NULL = no nulls. CONSTANT: only bit 0 is read. */
  const int32_t* indices; /* DICTIONARY: num_rows indices into values. */
  int32_t base_size;      /* DICTIONARY: number of base values; else 0. */
  int32_t mem;            /* vx355_mem of values / nulls / indices. VX355_MEM_DEVICE buffers must
                             be complete when the library's kernels read them: either they
                             are complete when the call is made, or the consuming context
                             was told to wait for the producer (vx355_stream_wait_event). */
} vx355_column;

/* RowVector (vector/ComplexVector.h) reduced to its decoded children. */
typedef struct vx355_batch {
  int32_t num_rows; /* vector_size_t: <= 2^31-1 */
  int32_t num_cols;
  const vx355_column* cols; /* host array */
} vx355_batch;

/* One caller-allocated flat output column. VARCHAR / VARBINARY values are 16-byte
 * StringViews; strings longer than 12 bytes (grouping keys of the generic hash mode, min / max
 * over strings) are non-inline views: in a VX355_MEM_DEVICE column they point into the operator's HBM arena
 * (valid while the handle lives), in a VX355_MEM_HOST column into a host buffer the handle
 * keeps until its next get_output — the shim copies or wraps them like any string buffer
 * (FlatVector::stringBuffers_). */
typedef struct vx355_out_column {
  int32_t type_kind;
  int32_t mem;      /* vx355_mem of values / nulls */
  void* values;     /* capacity rows (BOOLEAN: bit packed) */
  uint64_t* nulls;  /* capacity bits, 1 = valid; may be NULL when the caller
                       knows the column cannot be null (keys with
                       ignore_null_keys, count) */
} vx355_out_column;

/* ---- runtime ----------------------------------------------------------- */

/* Prepares one GPU for use by this process: HBM block cache, pinned-buffer
 * cache and the device's default execution context (one HIP stream + one pinned
 * mailbox). May be called for several devices: ONE process can drive all the
 * GPUs of a node; the first device initialised is the
 * process default, and the calling thread is bound to the device it just
 * initialised. Idempotent.
 *
 * Threading (exec/Driver.cpp:538, SURVEY.md section 8(b) "Threading"): there is
 * no library-wide lock. Every operator handle (vx355_agg, vx355_join_build,
 * vx355_join_probe) owns its own execution context — created on the calling
 * thread's device by *_create — so operators of different Drivers run
 * concurrently on different streams; one handle must not be used by two
 * threads at once (a Driver never does). Handle-less entry points (the
 * standalone kernels, memcpy helpers, join-table filter queries) run on the
 * default context of the calling thread's device and serialise on it. When an
 * entry point returns, everything it queued on its stream has completed. */
int vx355_init(int device);
/* Binds the calling thread to an initialised device: handles created afterwards
 * and handle-less calls use it. Threads that never call it use the process
 * default device. */
int vx355_set_device(int device);
int vx355_current_device(void); /* -1 before vx355_init */
void vx355_shutdown(void);
int vx355_abi_version(void);
int vx355_device_count(void);
const char* vx355_last_error(void);

/* Device memory helpers for callers that keep columns resident in HBM
 * (tests, bench, the adapter's staging buffers). */
void* vx355_device_malloc(size_t bytes);
void vx355_device_free(void* p);
int vx355_memcpy_h2d(void* dst, const void* src, size_t bytes);
int vx355_memcpy_d2h(void* dst, const void* src, size_t bytes);
int vx355_memset_d(void* dst, int value, size_t bytes);
/* Waits for every context (default + operator handles) of the calling thread's device. */
int vx355_synchronize(void);

/* Stream handshake with the producer of device-resident inputs. VX355_MEM_DEVICE
 * buffers must be complete before the library's kernels read them; instead of
 * synchronising the device the caller records a hipEvent_t on its own stream
 * after the producing kernel and makes the consuming context wait for it:
 *   hipEventRecord(ev, producer_stream);
 *   vx355_stream_wait_event(vx355_agg_stream(op), ev);   // or any *_stream()
 *   vx355_agg_add_input(op, &device_batch);
 * 'stream' is a hipStream_t returned by vx355_default_stream() / *_stream(),
 * 'event' a hipEvent_t (both passed as void* to keep HIP types out of the ABI). */
int vx355_stream_wait_event(void* stream, void* event);
void* vx355_default_stream(void); /* default context of the calling thread's device */

/* Per-kernel timing with HIP events on the library's own stream. While
 * enabled every launch of a profiled kernel is bracketed by two events;
 * vx355_profile_get sums the elapsed times after synchronising. */
int vx355_profile_enable(int on);
int vx355_profile_reset(void);
int vx355_profile_get(const char* kernel, double* total_ms, int64_t* launches);
/* Writes up to cap names of kernels seen since the last reset, '\n' joined. */
int vx355_profile_names(char* buf, size_t cap);

/* ---- memory beyond the operator's share (ABI 9) -------------------------------------------------
 * The reference's operators allocate from a MemoryPool with a capacity; past it they reclaim - spill
 * (HashBuild::ensureTableFits / reclaim, exec/HashBuild.cpp:995,1314; HashAggregation::reclaim,
 * exec/HashAggregation.cpp:562; HashProbe::reclaim, exec/HashProbe.cpp:2182) - or the query fails with
 * "Exceeded memory pool capacity". This library never spills (288 GB of HBM per
 * GPU, and a join beyond that is repartitioned across GPUs with vx355_join_repartition); what it does
 * guarantee is the failing half of that contract: an operator whose next allocation would pass the
 * limit - this one, or the GPU's free HBM - returns VX355_ENOMEM from the entry point that needed the
 * memory (add_input, no_more_input / finish, get_output), the message names the sizes, the handle
 * stays destroyable (destroy releases everything it held; nothing of it leaks into the block cache's
 * accounting) and other operators keep working. The adapter reports that as the Task's failure
 * (canReclaim() == false).
 * vx355_set_memory_limit: cap, in bytes, on the HBM blocks all operators of the calling thread's GPU
 * hold at one time (tables, staged input, scratch; not caller-owned buffers); 0 = no cap (default).
 * vx355_memory_usage: bytes held now, the peak since the previous call of this function, and bytes
 * parked in the block cache for reuse (not counted against the limit). Any pointer may be NULL. */
int vx355_set_memory_limit(int64_t bytes);
int vx355_memory_usage(int64_t* in_use, int64_t* peak, int64_t* cached);

/* What this GPU's HBM delivers to plain streaming kernels (measurement aid; bench.py quotes every
 * roofline fraction against the 8 TB/s datasheet peak and reports these next to it).
 * VX355_CEILING_READ: a read-only stream of 'bytes' bytes (16-byte nontemporal loads, nothing written);
 * VX355_CEILING_COPY: 'bytes' bytes read and as many written. Runs the kernel 'iterations' times on the
 * default context's stream (after two warm-up launches) between two HIP events and reports
 * bytes moved / elapsed time in GB/s. bytes >= 1 MiB. */
#define VX355_CEILING_READ 0
#define VX355_CEILING_COPY 1
/* VX355_CEILING_READ_COLUMNS: 'bytes' bytes as SEVEN column streams read side by side in the row -> lane
 * mapping of the fused aggregation kernel on TPC-H Q1's scan (two 16-byte view columns of which 8 bytes per
 * row are loaded, one 4-byte and four 8-byte columns: 68 bytes of HBM traffic per row), nothing computed:
 * what that kernel's arithmetic competes with on this box (interleaved streams deliver less than one). */
#define VX355_CEILING_READ_COLUMNS 2
int vx355_hbm_ceiling(int32_t kind, size_t bytes, int32_t iterations, double* gbytes_per_second);

/* ---- standalone kernels (parity-test surface) --------------------------- */

/* VectorHasher::hash (exec/VectorHasher.cpp:567-584, hashValues :86-126) for
 * n_keys columns in one pass: out[row] = h(key0) then hashMix(out[row],
 * h(key_i)) (common/base/BitUtil.h:775-784); nulls hash to kNullHash = 1.
 * rows: selection bitmap (SelectivityVector bits, 1 = selected) or NULL for
 * all rows; unselected slots of out are left untouched. mix_first != 0 mixes
 * key0 into the existing out[row] (the `mix` argument of the reference).
 * out/rows live in out_mem. */
int vx355_hash_columns(
    const vx355_batch* batch,
    const int32_t* key_cols,
    int32_t n_keys,
    const uint64_t* rows,
    int32_t mix_first,
    uint64_t* out,
    int32_t out_mem);

/* Per-key value-id mapping state: what VectorHasher holds after
 * enableValueRange (exec/VectorHasher.cpp:923-944). */
typedef struct vx355_value_id_spec {
  int64_t min;         /* min_ (inclusive, after reserve padding) */
  int64_t max;         /* max_ */
  uint64_t multiplier; /* multiplier_ */
} vx355_value_id_spec;

/* VectorHasher::computeValueIds (exec/VectorHasher.cpp:354-360; lookup = 0)
 * and VectorHasher::lookupValueIds (:550-565; lookup = 1) in range mode for
 * n_keys columns: result[row] = sum_i multiplier_i * (value_i - min_i + 1),
 * null contributes 0 (VectorHasher.h:560-566, VectorHasher.cpp:196-224). The
 * distinct-value mode of the same two functions is vx355_value_dict_* below.
 * lookup = 0: *all_mapped = 0 if any selected non-null value is out of range
 *   (the reference then re-decides the hash mode); result for such rows is
 *   unspecified. rows_out is not written.
 * lookup = 1: rows whose value is out of range are cleared in rows_out
 *   (proven misses); rows_out must hold ceil(num_rows/64) words and receives
 *   rows (or all-ones) AND mapped. */
int vx355_value_ids(
    const vx355_batch* batch,
    const int32_t* key_cols,
    const vx355_value_id_spec* specs,
    int32_t n_keys,
    const uint64_t* rows,
    int32_t lookup,
    uint64_t* result,
    uint64_t* rows_out,
    int32_t* all_mapped,
    int32_t out_mem);

/* VectorHasher in distinct-value mode (enableValueIds, exec/VectorHasher.cpp:906-921;
 * makeValueIds :128-161,196-224; valueId, VectorHasher.h:567-580): the value ids of a key column
 * are the 1-based insertion numbers of its distinct values, handed out in row order. The
 * dictionary lives in HBM and persists across batches like uniqueValues_.
 *  - compute (computeValueIds): result[row] = id (multiplier 1; null rows get 0) or result[row] +
 *    multiplier * id for the selected rows; new values are added in first-occurrence order.
 *    *all_mapped = 0 once the dictionary holds range_size values or more (the value that fills it
 *    is unmappable, VectorHasher.h:573-576): the caller re-decides the hash mode, as in range mode.
 *  - lookup (lookupValueIds, VectorHasher.cpp:408-492): values not in the dictionary are proven
 *    misses: their rows are cleared in rows_out (ceil(num_rows / 64) words; may be NULL).
 * Integer-like kinds, DATE, and strings of at most 7 bytes (their stringAsNumber image);
 * result / rows / rows_out live in 'mem'. */
typedef struct vx355_value_dict vx355_value_dict;
int vx355_value_dict_create(int32_t type_kind, int64_t range_size, vx355_value_dict** out);
int vx355_value_dict_compute(
    vx355_value_dict* d,
    const vx355_batch* batch,
    int32_t col,
    const uint64_t* rows,
    uint64_t multiplier,
    uint64_t* result,
    int32_t* all_mapped,
    int32_t mem);
int vx355_value_dict_lookup(
    vx355_value_dict* d,
    const vx355_batch* batch,
    int32_t col,
    const uint64_t* rows,
    uint64_t multiplier,
    uint64_t* result,
    uint64_t* rows_out,
    int32_t mem);
int64_t vx355_value_dict_size(const vx355_value_dict* d);
void vx355_value_dict_destroy(vx355_value_dict* d);

/* processFilterResults, flat case (exec/OperatorUtils.cpp:231-257): selected =
 * values & nulls & rows; idx_out receives the ascending row numbers of the set
 * bits (== FilterEvalCtx::selectedIndices), *n_out (a host int) their count.
 * nulls and rows may be NULL. The bitmaps and idx_out live in mem. */
int vx355_filter_compact(
    const uint64_t* values,
    const uint64_t* nulls,
    const uint64_t* rows,
    int32_t num_rows,
    int32_t* idx_out,
    int32_t* n_out,
    int32_t mem);

typedef enum vx355_partition_kind {
  VX355_PART_MODULO = 0,     /* hash % num_partitions (HashPartitionFunction.cpp:112-115) */
  VX355_PART_BIT_RANGE = 1,  /* (hash >> begin) & mask (HashBitRange.h:38-41) */
  VX355_PART_LOCAL_MODULO = 2,    /* XXH32(reverseBits(hash32)) % n (:25-30,:104-107) */
  VX355_PART_LOCAL_BIT_RANGE = 3  /* same then bit range (:96-99) */
} vx355_partition_kind;

/* HashPartitionFunction::partition (exec/HashPartitionFunction.cpp:76-118)
 * applied to precomputed VectorHasher hashes. For the BIT_RANGE kinds
 * num_partitions is ignored and bits [bit_begin, bit_end) are used. */
int vx355_partition(
    const uint64_t* hashes,
    int32_t num_rows,
    int32_t kind,
    int32_t num_partitions,
    int32_t bit_begin,
    int32_t bit_end,
    uint32_t* partitions_out,
    int32_t mem);

/* Dictionary over a dictionary (exec/OperatorUtils.cpp:393-422, wrapChild on an already wrapped
 * vector - what HashProbe::fillOutput produces over FilterProject's output, and what the probe of
 * TPC-H Q3's second join feeds the next operator): out[i] = inner[outer[i]] for i < num_rows.
 * inner has inner_size entries; an outer index outside [0, inner_size) is VX355_EINVAL. All three
 * arrays live in mem. */
int vx355_compose_indices(const int32_t* inner, int32_t inner_size, const int32_t* outer, int32_t num_rows,
                          int32_t* out, int32_t mem);

/* Repartitioning (exec/PartitionedOutput.cpp + exec/HashPartitionFunction.cpp,
 * what feeds an Exchange): reorders num_cols fixed-width columns so that the
 * rows of partition 0 come first, then partition 1, ... with the input order
 * kept inside each partition (stable), and reports the row count of every
 * partition. partitions[] is the output of vx355_partition. widths[c] is the
 * byte width of column c (1, 2, 4, 8 or 16). counts_out is a host array of
 * num_partitions entries; everything else lives in mem. num_partitions <= 64
 * (one partition per GPU of a node, or per spill bucket). */
int vx355_partition_scatter(
    const uint32_t* partitions,
    int32_t num_rows,
    int32_t num_partitions,
    const void* const* cols_in,
    const int32_t* widths,
    int32_t num_cols,
    void* const* cols_out,
    int64_t* counts_out,
    int32_t mem);

/* ---- PrestoPage wire format (serializers/PrestoSerializer.h) --------------- */

typedef enum vx355_page_flags {
  /* A PrestoOutputStreamListener is attached (serializers/PrestoSerializerSerializationUtils.h:
   * 167-187): codec marker bit 4 and the CRC32 of [column count + columns | codec | numRows |
   * uncompressedSize] in the header. Host output only. Without it codec and checksum are 0. */
  VX355_PAGE_CHECKSUM = 1,
  /* PrestoOptions::useLosslessTimestamp: {seconds, nanos} instead of milliseconds. */
  VX355_PAGE_LOSSLESS_TIMESTAMP = 2
  /* bits 8..15: PrestoOptions::compressionKind of the exchange, see VX355_PAGE_COMPRESSION (ABI 9) */
} vx355_page_flags;

/* common::CompressionKind (common/compression/Compression.h:28-37), the values Velox itself uses. The
 * page body of a compressed PrestoPage is what the folly codec of that kind emits
 * (compressionKindToCodec, common/compression/Compression.cpp:27-46): ZLIB an RFC 1950 stream, SNAPPY
 * raw snappy, ZSTD a zstd frame, LZ4 ONE raw LZ4 block (folly's CodecType::LZ4 does not store the
 * length: the page header's uncompressedSize is it), GZIP an RFC 1952 member. LZO and LZ4_HADOOP have
 * no folly codec ("Not support ... in folly") and are VX355_EUNSUPPORTED here too. LZ4 and SNAPPY are
 * coded by the library itself; ZSTD and ZLIB / GZIP need libzstd.so.1 / libz.so.1 on the host (loaded
 * on first use; without them only those kinds are VX355_EUNSUPPORTED). */
typedef enum vx355_compression_kind {
  VX355_COMPRESSION_NONE = 0,
  VX355_COMPRESSION_ZLIB = 1,
  VX355_COMPRESSION_SNAPPY = 2,
  VX355_COMPRESSION_LZO = 3,
  VX355_COMPRESSION_ZSTD = 4,
  VX355_COMPRESSION_LZ4 = 5,
  VX355_COMPRESSION_GZIP = 6
} vx355_compression_kind;
/* The exchange's compression kind inside a vx355_page_flags word (vx355_presto_deserialize). A page says
 * only THAT it is compressed (codec marker bit 1, PrestoSerializerSerializationUtils.h:37-45); which
 * codec is the reader's configuration, as in the reference (PrestoSerializer.cpp:144-145). */
#define VX355_PAGE_COMPRESSION(kind) (((kind) & 0xff) << 8)
#define VX355_PAGE_COMPRESSION_OF(flags) (((flags) >> 8) & 0xff)

/* What PartitionedOutput's Destination does with the rows routed to it
 * (exec/PartitionedOutput.cpp:59-133: IterativeVectorSerializer::append(rows) + flush), for all
 * destinations of a batch in one call: page p holds the batch rows
 * rows[offsets[p] .. offsets[p + 1]) in that order (rows == NULL: the batch rows themselves in
 * that range), every column flattened, uncompressed:
 *   numRows i32 | codec i8 | uncompressedSize i32 | size i32 | checksum i64 | numColumns i32 |
 *   per column: encoding name (i32 length + BYTE_ARRAY / SHORT_ARRAY / INT_ARRAY / LONG_ARRAY /
 *   VARIABLE_WIDTH, PrestoSerializerSerializationUtils.cpp:997-1040) | numRows i32 |
 *   [VARIABLE_WIDTH: i32 end offset per row] | hasNulls i8 [| null bits, first row in the most
 *   significant bit, 1 = null] | [VARIABLE_WIDTH: total bytes i32] | the non-null values
 *   (VectorStream::flush, serializers/VectorStream.cpp:207-299).
 * BOOLEAN travels as one byte per value, TIMESTAMP as milliseconds (Timestamp::toMillis; out
 * of range -> VX355_EUSER) unless VX355_PAGE_LOSSLESS_TIMESTAMP. A range without rows yields
 * no bytes (Destination::flush returns early). offsets / page_offsets are host arrays of
 * num_pages + 1 entries; rows lives in rows_mem, out in out_mem. out == NULL only fills
 * page_offsets (the exact sizes: page p occupies [page_offsets[p], page_offsets[p + 1]) of out),
 * so the caller can allocate and call again.
 * A VX355_ROW column (see vx355_type_kind) is written in the ROW encoding (VectorStream::flush,
 * serializers/VectorStream.cpp:236-262; serializeRowVector, PrestoSerializerSerializationUtils.cpp:
 * 883-919): "ROW" | number of fields i32 | the fields' column streams, each holding only the rows
 * whose struct is NOT null | numRows i32 | numRows + 1 offsets i32 (0, then + 1 behind every
 * non-null struct) | hasNulls i8 [| null bits]. This is how a PARTIAL avg travels to a stock Velox
 * FINAL aggregation: ROW(DOUBLE sum, BIGINT count). */
int vx355_presto_serialize(
    const vx355_batch* batch,
    const int32_t* rows,
    int32_t rows_mem,
    const int64_t* offsets,
    int32_t num_pages,
    int32_t flags,
    void* out,
    int64_t out_capacity,
    int32_t out_mem,
    int64_t* page_offsets);

/* The writer's last step for a compressing exchange (flushCompressed,
 * serializers/PrestoSerializerSerializationUtils.h:279-334), as a pure host function (no GPU, no
 * vx355_init needed): 'page' is one uncompressed page as vx355_presto_serialize wrote it to host
 * memory; its body (column count + columns) is compressed with the codec of 'compression'. When the
 * result is larger than uncompressedSize x min_ratio (PrestoOptions::minCompressionRatio, 0.8 by
 * default) the page is emitted as it came, exactly as the reference does; otherwise the header gets
 * the compressed bit, size = compressed bytes, and - when the page carries a checksum - the CRC32
 * over [compressed body | codec | numRows | uncompressedSize] (PrestoSerializer.cpp:39-79). The output
 * never exceeds the input: out_capacity >= size always suffices. *out_size = bytes written. */
int vx355_presto_compress_page(const void* page, int64_t size, int32_t compression, float min_ratio, void* out,
                               int64_t out_capacity, int64_t* out_size);
/* The inverse, also pure host work: a page with the compressed bit becomes the uncompressed page a
 * non-compressing writer would have sent (checksum verified over the compressed bytes first and
 * re-computed for the new body); a page without the bit is copied. out_capacity >= 21 + the header's
 * uncompressedSize (little-endian int32 at byte 5 of the page). vx355_presto_deserialize does this by
 * itself for every compressed page it is given. */
int vx355_presto_uncompress_page(const void* page, int64_t size, int32_t compression, void* out, int64_t out_capacity,
                                 int64_t* out_size);

/* The other direction, what an Exchange does with the pages it received
 * (PrestoVectorSerde::deserialize, serializers/PrestoSerializer.cpp:120-200, appending page
 * after page into one RowVector): pages (host memory; a checksum is verified when
 * the codec marker carries one, "Received corrupted serialized page." -> VX355_EUSER) become
 * flat columns in HBM, rows of page 0 first (RLE and DICTIONARY columns are flattened on the
 * way). Compressed pages (codec marker bit 1; ABI 9) are uncompressed on the host with the codec
 * named by VX355_PAGE_COMPRESSION(kind) in 'flags' (PrestoSerializer.cpp:185-199) - a compressed
 * page without a kind in the flags is VX355_EINVAL, a body the codec rejects VX355_EUSER - and
 * device_bytes must then hold 21 + uncompressedSize bytes for such a page instead of its wire size;
 * encrypted pages stay VX355_EUNSUPPORTED. types[] is the RowType the exchange expects; a
 * page whose column encodings do not fit it is a VX355_EUSER. The page bytes are copied into
 * device_bytes (>= the sum of sizes); views of strings longer than 12 bytes point into that
 * buffer, so the caller keeps it as long as the columns (a vector's string buffer). cols: device
 * memory, capacity_rows rows each (the row count of a page is its first little-endian int32,
 * so the caller sizes them before the call). Null rows hold the type's default value.
 * Struct columns: types[] / cols[] list the column tree in prefix order - a struct's own entry is
 * VX355_ROW_OF(number of fields) in types[] and a vx355_out_column of kind VX355_ROW in cols[]
 * (only its nulls buffer is written: the struct's validity), followed by one entry per field; a
 * field's column has a row per struct row, null where the struct is null (readRowVector,
 * PrestoSerializerDeserializationUtils.cpp:1041-1110). num_cols counts the entries; the pages'
 * own column count is the number of top-level entries. */
#define VX355_ROW_OF(num_fields) (VX355_ROW | ((num_fields) << 8))
int vx355_presto_deserialize(
    const void* const* pages,
    const int64_t* sizes,
    int32_t num_pages,
    const int32_t* types,
    int32_t num_cols,
    int32_t flags,
    void* device_bytes,
    int64_t device_bytes_capacity,
    vx355_out_column* cols,
    int64_t capacity_rows,
    int64_t* rows_out);

/* ---- FilterProject for the TPC-H Q1 / Q3 expression class ----------------- */

/* The step immediately upstream of HashAggregation / HashProbe
 * (exec/FilterProject.cpp:102-275): a conjunction of column-vs-constant
 * comparisons (nulls fail, like exec/OperatorUtils.cpp:231-257) followed by
 * projections of the form f0 * f1 * ... with f_i = scale_i * column_i +
 * offset_i, evaluated left to right in DOUBLE (REAL and integer columns are
 * widened first). That covers l_shipdate <= DATE c, c_mktsegment = 'BUILDING',
 * l_extendedprice * (1 - l_discount) * (1 + l_tax) (exec/tests/utils/
 * TpchQueryBuilder.cpp:203-252,467-558); anything else stays on the CPU
 * evaluator. */
typedef enum vx355_cmp {
  VX355_CMP_EQ = 0,
  VX355_CMP_NE = 1,
  VX355_CMP_LT = 2,
  VX355_CMP_LE = 3,
  VX355_CMP_GT = 4,
  VX355_CMP_GE = 5
} vx355_cmp;

typedef struct vx355_filter_term {
  int32_t col;        /* batch column */
  int32_t cmp;        /* vx355_cmp */
  int32_t const_kind; /* VX355_BIGINT: i64 (any integer-like or DATE column);
                         VX355_DOUBLE: f64 (REAL / DOUBLE column);
                         VX355_VARCHAR: str (EQ / NE only, <= 12 bytes) */
  int32_t str_size;
  int64_t i64;
  double f64;
  char str[16];
} vx355_filter_term;

typedef struct vx355_factor {
  int32_t col; /* -1: the factor is the constant 'offset' */
  int32_t pad;
  double scale;
  double offset;
} vx355_factor;

typedef struct vx355_projection {
  int32_t num_factors; /* 1..4 */
  int32_t pad;
  vx355_factor factors[4];
} vx355_projection;

/* Evaluates the filter over all rows of 'batch', writes the ascending selected
 * row numbers to idx_out (capacity num_rows; == FilterEvalCtx::selectedIndices,
 * the indices FilterProject wraps pass-through columns with,
 * exec/OperatorUtils.cpp:393-422) and, for each projection j, the DOUBLE
 * results of the selected rows to proj_out[j] (capacity num_rows, flat) with
 * validity in proj_nulls_out[j] (may be NULL when no input can be null; a
 * null input makes the result null). n_terms == 0 selects every row.
 * idx_out / proj_out / proj_nulls_out live in out_mem; *n_out is a host int. */
int vx355_filter_project(
    const vx355_batch* batch,
    const vx355_filter_term* terms,
    int32_t n_terms,
    const vx355_projection* projections,
    int32_t n_projections,
    int32_t* idx_out,
    int32_t* n_out,
    double* const* proj_out,
    uint64_t* const* proj_nulls_out,
    int32_t out_mem);

/* ---- HashAggregation (exec/HashAggregation.h, exec/GroupingSet.h) ------- */

typedef enum vx355_agg_kind {
  VX355_AGG_SUM = 0,        /* SumAggregate.cpp:39-118 */
  VX355_AGG_COUNT = 1,      /* count(x), CountAggregate.cpp:27-147; x of any column type (raw input) */
  VX355_AGG_COUNT_STAR = 2, /* count(*) */
  VX355_AGG_MIN = 3,        /* MinMaxAggregateBase.cpp:101-305; over VARCHAR / VARBINARY input
                               :305-480 (any step: the intermediate type is the input type).
                               Strings longer than 12 bytes come out as
                               described at vx355_out_column. */
  VX355_AGG_MAX = 4,
  VX355_AGG_AVG = 5         /* AverageAggregateBase.h:66-260 */
} vx355_agg_kind;

/* core::AggregationNode::Step (core/PlanNode.h:1122-1131), same values. */
typedef enum vx355_agg_step {
  VX355_STEP_PARTIAL = 0,      /* raw in, intermediate out */
  VX355_STEP_FINAL = 1,        /* intermediate in, final out */
  VX355_STEP_INTERMEDIATE = 2, /* intermediate in, intermediate out */
  VX355_STEP_SINGLE = 3        /* raw in, final out */
} vx355_agg_step;

typedef struct vx355_agg_fn {
  int32_t kind;       /* vx355_agg_kind */
  int32_t input_col;  /* batch column; -1 for count(*) */
  int32_t input_col2; /* avg with intermediate input: the BIGINT count child
                         of ROW(DOUBLE sum, BIGINT count); else -1 */
  int32_t input_type; /* raw input vx355_type_kind (decides result type even
                         for intermediate input: sum(REAL) returns REAL) */
  int32_t mask_col;   /* BOOLEAN column, FILTER (WHERE m); -1 = none
                         (exec/AggregationMasks.h) */
  int32_t flags;      /* vx355_agg_fn_flags */
} vx355_agg_fn;

typedef enum vx355_agg_fn_flags {
  /* agg(DISTINCT x) (core::AggregationNode::Aggregate::distinct, exec/DistinctAggregations.cpp):
   * the function sees each distinct input value of a group once, nulls included (and then
   * ignores them as usual); the mask applies before the de-duplication
   * (GroupingSet.cpp:317-332). SINGLE step only, like the reference ("Partial aggregations
   * over distinct inputs are not supported", GroupingSet.cpp:117-121). For min / max the flag
   * changes nothing. ORDER BY inside an aggregate needs no flag: sum / count / min / max / avg
   * are registered orderSensitive = false and the reference drops their sorting keys itself
   * (exec/AggregateInfo.cpp:124-138), so the shim passes such aggregates without the keys. */
  VX355_AGG_FN_DISTINCT = 1
} vx355_agg_fn_flags;

typedef struct vx355_agg_spec {
  int32_t num_keys; /* 0 = global aggregation: always one output row */
  const int32_t* key_cols;
  const int32_t* key_types;
  int32_t num_aggs;
  const vx355_agg_fn* aggs;
  int32_t step;             /* vx355_agg_step */
  int32_t ignore_null_keys; /* AggregationNode::ignoreNullKeys */
  int32_t flags;            /* vx355_agg_flags */
  int32_t pad;
} vx355_agg_spec;

typedef enum vx355_agg_flags {
  /* The consumer does not depend on the group order (a FINAL aggregation, an exchange, an
   * ORDER BY above): groups come out in table order instead of first-seen order
   * (GroupingSet.cpp:828-839), which saves the sort of the groups by first input row and
   * lets the radix passes of a high-cardinality aggregation drop the row number from their
   * records (12 instead of 16 bytes over a direct-index table, 16 instead of 24 over an
   * open-addressing one): a quarter to a third of the time of a 100 M-group aggregation.
   * Results per group are unchanged. */
  VX355_AGG_UNORDERED_OUTPUT = 1
} vx355_agg_flags;

typedef struct vx355_agg vx355_agg;

/* HashAggregation::initialize (exec/HashAggregation.cpp:44-130). */
int vx355_agg_create(const vx355_agg_spec* spec, vx355_agg** out);
/* Operator fusion FilterProject -> HashAggregation (the adapter replaces both
 * operators with one, like experimental/cudf/exec/ToCudf.cpp:155-213 does for
 * its own fusions). After this call the batches handed to vx355_agg_add_input
 * are the FilterProject INPUT batches: rows failing the filter are skipped and
 * an aggregate whose input_col is VX355_PROJECTION_COL_BASE + j reads
 * projection j. Results equal vx355_filter_project followed by
 * vx355_agg_add_input on its output. Call before the first add_input; raw
 * steps (partial / single) only; projection inputs are DOUBLE. */
#define VX355_PROJECTION_COL_BASE (1 << 20)
int vx355_agg_set_fused_input(
    vx355_agg* h,
    const vx355_filter_term* terms,
    int32_t n_terms,
    const vx355_projection* projections,
    int32_t n_projections);
/* HashAggregation::addInput (:191-236) -> GroupingSet::addInput
 * (exec/GroupingSet.cpp:190-223,288-365). Host batches of fewer than 256 K rows
 * (VX355_AGG_COALESCE_ROWS) are copied into host-side column buffers and
 * aggregated in larger pieces; an error caused by such rows (integer overflow,
 * unsupported value) is reported by the call that flushes them: a later
 * add_input, no_more_input or get_output. */
int vx355_agg_add_input(vx355_agg* h, const vx355_batch* batch);
/* Operator::noMoreInput (exec/Operator.h:252). */
int vx355_agg_no_more_input(vx355_agg* h);
/* Number of output columns: keys, then per aggregate 1 column (2 for avg in
 * partial/intermediate steps: DOUBLE sum, BIGINT count), and their types. */
int vx355_agg_output_types(const vx355_agg* h, int32_t* types, int32_t cap, int32_t* n);
/* HashAggregation::getOutput (:357-424) -> GroupingSet::getOutput (:810-884):
 * groups in first-seen order, at most max_rows per call. Valid after
 * no_more_input. *finished = 1 once every group has been returned. */
int vx355_agg_get_output(
    vx355_agg* h,
    vx355_out_column* cols,
    int32_t num_cols,
    int32_t max_rows,
    int32_t* n_out,
    int32_t* finished);

/* ---- asynchronous boundary (ABI 5) -------------------------------------------------------
 * exec::Operator::isBlocked(ContinueFuture*) / needsInput() (exec/Operator.h:280-299): a Driver
 * thread must not sit in addInput while staging copies, transfers and kernels run.
 * vx355_agg_add_input_async queues the batch for the handle's worker thread and returns at once
 * with a ticket (1, 2, ...); batches are processed in submission order. The buffers 'batch' points
 * to must stay valid until its ticket has completed (the descriptor itself is copied): the shim
 * keeps the RowVectorPtr and drops it when vx355_agg_poll reports completed >= ticket.
 * vx355_agg_poll never blocks: submitted / completed ticket counts (needsInput = few in flight,
 * isBlocked = completed < submitted when the shim wants to wait). vx355_agg_wait blocks until the
 * queue is empty and returns the first failure among the queued batches (its message is then the
 * calling thread's vx355_last_error(); batches behind a failed one are skipped). The failure stays
 * with the handle: every later wait, add_input(_async), no_more_input, get_output, flush ... returns
 * it again until the handle is destroyed - the operator is short of input and must not produce a
 * result quietly. (get_stats and poll stay usable.)
 * Every other entry point of the handle (add_input, no_more_input, get_output, flush, get_stats,
 * destroy ...) waits for the queue first, so mixing synchronous and asynchronous calls is safe and
 * ordered. The same three calls exist for HashBuild. */
int vx355_agg_add_input_async(vx355_agg* h, const vx355_batch* batch, int64_t* ticket_out);
int vx355_agg_poll(vx355_agg* h, int64_t* submitted, int64_t* completed);
int vx355_agg_wait(vx355_agg* h);

/* Queued no_more_input / get_output (ABI 7): exec::Operator::noMoreInput() and getOutput() must not make
 * the Driver thread sit through the last queued batches, the listing of the groups and the copies into
 * the result vectors either (exec/Driver.cpp:538-800: a blocked operator hands the thread back).
 * vx355_agg_no_more_input_async queues noMoreInput behind the batches submitted so far.
 * vx355_agg_get_output_async queues ONE page of output (the arguments of vx355_agg_get_output; the
 * descriptors are copied, the buffers they point to must stay valid until the ticket completes); when the
 * page is there - or was skipped behind a failure - 'done' (may be NULL) is called ON THE LIBRARY'S WORKER
 * THREAD with (done_arg, status, num_rows, finished): fulfil the ContinuePromise behind the future
 * isBlocked() returned, nothing heavier. The callback has RETURNED by the time vx355_agg_poll / _wait report
 * completed >= ticket (ABI 8: it runs right before the ticket counts as completed), so done_arg may be freed
 * from then on; vx355_agg_output_result may also be called from inside or right after the callback - it hands
 * the page's (num_rows, finished) or its failure to the Driver thread, once, and is VX355_EINVAL only for a
 * page that is neither filled nor skipped yet. Several pages may be queued; they are filled in order. Every
 * synchronous entry point still waits for the queue first. */
typedef void (*vx355_output_done_fn)(void* arg, int status, int32_t num_rows, int32_t finished);
int vx355_agg_no_more_input_async(vx355_agg* h, int64_t* ticket_out);
int vx355_agg_get_output_async(vx355_agg* h, const vx355_out_column* cols, int32_t num_cols, int32_t max_rows,
                               vx355_output_done_fn done, void* done_arg, int64_t* ticket_out);
int vx355_agg_output_result(vx355_agg* h, int64_t ticket, int32_t* num_rows, int32_t* finished);

/* hashtable.* runtime stats (exec/HashTable.h:155-182). */
typedef struct vx355_agg_stats {
  int64_t num_groups;   /* hashtable.numDistinct */
  int64_t capacity;     /* hashtable.capacity */
  int64_t num_rehashes; /* hashtable.numRehashes */
  int32_t hash_mode;    /* 0 kHash, 1 kArray, 2 kNormalizedKey (BaseHashTable::HashMode) */
  int32_t reserved;     /* launches of a hiprtc-instantiated shape-specialised kernel */
  int64_t input_rows;
  int64_t deferred_rows; /* rows replayed after a key-range widening */
  int64_t radix_launches; /* chunks aggregated through the radix-partitioned LDS path */
  int64_t table_bytes;    /* HBM held by the group table (what isPartialFull compares with
                             max_partial_aggregation_memory, GroupingSet::isPartialFull) */
  int64_t num_flushes;    /* vx355_agg_flush calls completed (kFlushTimes) */
  int64_t compact_record_launches; /* radix launches whose passes moved records without row number and mask
                             (VX355_AGG_UNORDERED_OUTPUT, one flat operand): 12 bytes {key : 32, operand} over
                             a direct-index table, 16 bytes {key, operand} over an open-addressing one - ABI 7 */
} vx355_agg_stats;
int vx355_agg_get_stats(const vx355_agg* h, vx355_agg_stats* out);

/* What an operator cost on the GPU side so far (ABI 8) - the gpu.* runtime stats the adapter reports
 * next to the reference's hashtable.* ones ("Metrics"; exec/Operator.h:359-362
 * addRuntimeStat). Counted per operator handle (its execution context), no profiling mode needed,
 * readable at any time without waiting for queued work:
 *   busy_nanos   nanoseconds between entering an entry point of the handle and its stream being
 *                drained (every entry point returns with its work complete): the kernels, copies and
 *                host-side launch work of the operator                        -> gpu.kernelNanos
 *   h2d_bytes    bytes copied host -> HBM (staging of host vectors)           -> gpu.h2dBytes
 *   d2h_bytes    bytes moved HBM -> host (output pages, small read-backs; copies and the
 *                stores of kernels that write a pinned page or the mailbox themselves) -> gpu.d2hBytes
 *   input_bytes  bytes of the input columns handed to kernels (values, null bitmaps, indices): what one
 *                pass over the input reads from HBM                           -> gpu.hbmBytesRead
 *   launches     kernels launched                                             -> gpu.kernelLaunches
 * gpu.hbmBytesWritten is reported by the adapter as d2h_bytes plus the operator's table bytes. */
typedef struct vx355_gpu_stats {
  int64_t busy_nanos;
  int64_t h2d_bytes;
  int64_t d2h_bytes;
  int64_t input_bytes;
  int64_t launches;
  int64_t reserved;
} vx355_gpu_stats;
int vx355_agg_get_gpu_stats(const vx355_agg* h, vx355_gpu_stats* out);
/* The two numbers the shim's isPartialFull / abandon checks need, WITHOUT waiting (ABI 7):
 * vx355_agg_get_stats drains the handle's queue and seals the open ingest chunk, which stalls the
 * Driver thread and defeats the asynchronous boundary when called per batch. table_bytes / num_groups
 * are as of the last batch the library finished feeding (see vx355_agg_poll for which one that is);
 * after a flush has been drained they are the empty table's. Either pointer may be NULL. */
int vx355_agg_table_bytes(const vx355_agg* h, int64_t* table_bytes, int64_t* num_groups);
/* Bytes the groups held NOW occupy (ABI 8): the table's allocation scaled by the share of its rows in use,
 * plus the DISTINCT sets. vx355_agg_table_bytes reports the allocation, which the library keeps across a
 * flush (as GroupingSet::resetTable(freeTable = false) keeps the reference's): after the first flush only
 * this number says how full the table is, and it is what the adapter compares with
 * max_partial_aggregation_memory every time (exec/HashAggregation.cpp:191-236). Never waits. */
int vx355_agg_bytes_in_use(const vx355_agg* h, int64_t* bytes);

/* Partial-aggregation flush (HashAggregation.cpp:191-236,293-327: when the partial table
 * is "full" the operator emits what it has and starts over). PARTIAL / INTERMEDIATE steps
 * with grouping keys only. After this call vx355_agg_get_output lists the groups
 * accumulated so far (first-seen order) although no_more_input has not been called; when it
 * reports finished the table is empty again (GroupingSet::resetTable) and add_input carries
 * on. WHEN to flush is the shim's policy, from vx355_agg_get_stats: table_bytes against
 * max_partial_aggregation_memory (16 MB default, core/QueryConfig.h), or the abandon test
 * num_groups / input_rows >= abandon_partial_aggregation_min_pct after
 * abandon_partial_aggregation_min_rows (HashAggregation.cpp:185-189). */
int vx355_agg_flush(vx355_agg* h);

/* Abandoned partial aggregation (HashAggregation.cpp:185-189,357-375 ->
 * GroupingSet::toIntermediate, exec/GroupingSet.cpp:1589-1675): the raw input rows of
 * 'batch' in the PARTIAL step's output layout WITHOUT grouping, one output row per input
 * row, as if every row were its own group: sum -> the value (BIGINT for integer inputs,
 * DOUBLE for REAL / DOUBLE) or null; count -> 1 or 0, never null; min / max -> the value;
 * avg -> (DOUBLE value, BIGINT 1) or (null, null). A false or null mask and a null input
 * make the row inactive for that aggregate. 'cols' receives the aggregate columns only, in
 * output order (the key columns pass through unchanged: the shim reuses the input
 * vectors), each with capacity batch->num_rows. Raw-input steps only; stateless. */
int vx355_agg_to_intermediate(vx355_agg* h, const vx355_batch* batch, vx355_out_column* cols, int32_t num_cols);
void vx355_agg_destroy(vx355_agg* h);
/* hipStream_t of the operator's execution context (see vx355_stream_wait_event). */
void* vx355_agg_stream(vx355_agg* h);

/* ---- HashBuild / HashProbe (exec/HashBuild.h, exec/HashProbe.h) --------- */

/* core::JoinType (core/PlanNode.h:3081-3165), same values. All twelve kinds run
 * on the device. Null-aware semantics (HashJoinNode::isNullAware) are supported
 * for ANTI and LEFT_SEMI_PROJECT, isNullAsValue keys (IS NOT DISTINCT FROM) for every kind;
 * null-aware RIGHT_SEMI_PROJECT returns VX355_EUNSUPPORTED at create. An extra join filter
 * (vx355_join_probe_set_filter) works with every kind, null aware or not, except the two counting
 * ones (the reference has none there either, HashProbe.cpp:1345-1365). Build and
 * probe must be created with the same join type. Counting joins keep one
 * remaining-count per distinct build key in the table: probe them from one
 * thread at a time. */
typedef enum vx355_join_type {
  VX355_JOIN_INNER = 0,
  VX355_JOIN_LEFT = 1,
  VX355_JOIN_RIGHT = 2,
  VX355_JOIN_FULL = 3,
  VX355_JOIN_LEFT_SEMI_FILTER = 4,
  VX355_JOIN_COUNTING_LEFT_SEMI_FILTER = 5,
  VX355_JOIN_LEFT_SEMI_PROJECT = 6,
  VX355_JOIN_RIGHT_SEMI_FILTER = 7,
  VX355_JOIN_RIGHT_SEMI_PROJECT = 8,
  VX355_JOIN_ANTI = 9,
  VX355_JOIN_COUNTING_ANTI = 10,
  VX355_JOIN_RIGHT_ANTI = 11
} vx355_join_type;

typedef struct vx355_join_build_spec {
  int32_t num_keys;
  const int32_t* key_cols;
  const int32_t* key_types;
  int32_t num_dependents; /* build-side payload columns */
  const int32_t* dependent_cols;
  const int32_t* dependent_types;
  int32_t join_type;  /* vx355_join_type */
  int32_t null_aware; /* HashJoinNode::isNullAware */
  int32_t null_as_value; /* HashJoinNode::isNullAsValue: keys compare IS NOT DISTINCT FROM (NULL equals
                            NULL), what INTERSECT / EXCEPT plan their counting joins with
                            (core/PlanNode.h:3442-3445); rows with null keys enter the table
                            (HashBuild.cpp:273,477) and null probe keys are looked up
                            (HashProbe.cpp:787). Excludes null_aware. */
  int32_t drop_duplicates; /* HashJoinNode::canDropDuplicates (core/PlanNode.h:3391-3398; HashBuild.cpp:517-548): a
                              left semi (filter / project) or anti join WITHOUT an extra filter only asks
                              whether a key exists; rows whose key is already in the table are not linked
                              (table statistics report unique keys, probes never walk a chain). Any other join
                              type with this flag, or vx355_join_probe_set_filter on such a table, is
                              VX355_EINVAL. Counting joins always keep one entry per key and its count. */
} vx355_join_build_spec;

typedef struct vx355_join_build vx355_join_build;
typedef struct vx355_join_table vx355_join_table;
typedef struct vx355_join_probe vx355_join_probe;

/* HashBuild::initialize / setupTable (exec/HashBuild.cpp:239-300). */
int vx355_join_build_create(const vx355_join_build_spec* spec, vx355_join_build** out);
/* HashBuild::addInput (:442-598): drops rows with a null key, appends keys and
 * dependents to the HBM-resident build columns. */
int vx355_join_build_add_input(vx355_join_build* h, const vx355_batch* batch);
/* Asynchronous forms, as for the aggregation (see vx355_agg_add_input_async): queue / poll / wait;
 * vx355_join_build_finish waits for this build's and the peers' queues. */
int vx355_join_build_add_input_async(vx355_join_build* h, const vx355_batch* batch, int64_t* ticket_out);
int vx355_join_build_poll(vx355_join_build* h, int64_t* submitted, int64_t* completed);
int vx355_join_build_wait(vx355_join_build* h);
/* HashBuild::noMoreInput -> finishHashBuild (:799-993) ->
 * HashTable::prepareJoinTable (exec/HashTable.cpp:1989-2069). others: build
 * handles of the peer Drivers whose rows are merged into this table (may be
 * NULL/0). The returned table carries one reference. */
int vx355_join_build_finish(
    vx355_join_build* h,
    vx355_join_build* const* others,
    int32_t num_others,
    vx355_join_table** out);
void vx355_join_build_destroy(vx355_join_build* h);
void* vx355_join_build_stream(vx355_join_build* h);

/* HashJoinBridge::setHashTable / tableOrFuture (exec/HashJoinBridge.h:57,116)
 * hand a shared_ptr; here: explicit reference counting. */
void vx355_join_table_retain(vx355_join_table* t);
void vx355_join_table_release(vx355_join_table* t);

typedef struct vx355_join_table_stats {
  int64_t num_rows;     /* build rows with non-null keys */
  int64_t num_distinct; /* distinct keys */
  int64_t capacity;     /* slots */
  int32_t hash_mode;    /* as vx355_agg_stats.hash_mode */
  int32_t has_duplicates;
} vx355_join_table_stats;
int vx355_join_table_get_stats(const vx355_join_table* t, vx355_join_table_stats* out);
/* gpu.* counters of a build / probe operator (see vx355_gpu_stats). */
int vx355_join_build_get_gpu_stats(const vx355_join_build* h, vx355_gpu_stats* out);
int vx355_join_probe_get_gpu_stats(const vx355_join_probe* h, vx355_gpu_stats* out);

/* ---- dynamic filters from the build side (HashProbe::pushdownDynamicFilters,
 * exec/HashProbe.cpp:408-457) --------------------------------------------------
 * After the table is finished the reference pushes one filter per join key to
 * the probe-side scan: VectorHasher::getFilter (exec/VectorHasher.cpp:731-780)
 * -> common::createBigintValues (type/Filter.cpp:1052-1114: BigintRange /
 * bitmask / hash table, decided from min, max and the value list) while the
 * key has at most kMaxDistinct = 100'000 distinct values
 * (exec/VectorHasher.h:139), else a BigintValuesUsingBloomFilter
 * (type/Filter.h:1294-1360: SplitBlockBloomFilter over folly::hasher<int64_t>,
 * sized by numBlocks(table.numDistinct, 0.01); HashTable.cpp:1133-1188).
 * The shim asks for the description, then for the value list or the Bloom
 * blocks, and constructs the common::Filter on the Velox side. Integer key
 * kinds only (TINYINT..BIGINT, DATE as INTEGER); other kinds report NONE. */
enum vx355_key_filter_kind {
  VX355_KEY_FILTER_NONE = 0,
  VX355_KEY_FILTER_VALUES = 1, /* <= 100'000 distinct values: fetch them with _values */
  VX355_KEY_FILTER_BLOOM = 2   /* more: fetch Bloom blocks with _bloom */
};
typedef struct vx355_key_filter {
  int32_t kind;          /* vx355_key_filter_kind */
  int32_t pad;
  int64_t min, max;      /* over the non-null build values of the key (kind != NONE) */
  int64_t num_distinct;  /* VALUES: exact count; BLOOM: the table's distinct-key count (the capacity the reference sizes with) */
} vx355_key_filter;
int vx355_join_table_key_filter(vx355_join_table* t, int32_t key, vx355_key_filter* out);
/* Ascending distinct values of the key; capacity >= num_distinct. */
int vx355_join_table_key_filter_values(vx355_join_table* t, int32_t key, int64_t* values_out, int64_t capacity,
                                       int32_t mem, int64_t* n_out);
/* SplitBlockBloomFilter::numBlocks (common/base/SplitBlockBloomFilter.cpp:27-34) for blocks of 'lanes'
 * 32-bit words: 8 = 256-bit SIMD hosts (AVX2), 4 = 128-bit (SSE / NEON). */
int64_t vx355_bloom_num_blocks(int64_t num_elements, double false_positive, int32_t lanes);
/* Fills num_blocks * lanes u32 words (zeroed first) with every non-null build value of the key:
 * bit-identical to inserting them into the reference's SplitBlockBloomFilter of that block width. */
int vx355_join_table_key_filter_bloom(vx355_join_table* t, int32_t key, int32_t lanes, uint32_t* blocks_out,
                                      int64_t num_blocks, int32_t mem);
/* BigintValuesUsingBloomFilter::testInt64 over a column (device-side scans): rows_out = rows (all if NULL)
 * AND value is not null AND mayContain(value). 'blocks' and the bitmaps live in 'mem'. */
int vx355_bloom_test(const uint32_t* blocks, int64_t num_blocks, int32_t lanes, const vx355_column* column,
                     int32_t num_rows, const uint64_t* rows, uint64_t* rows_out, int32_t mem);

typedef struct vx355_join_probe_spec {
  int32_t num_keys;
  const int32_t* key_cols; /* probe-side key columns */
  int32_t join_type;
  int32_t null_aware; /* ANTI (NOT IN) and LEFT_SEMI_PROJECT (IN as a column); with an extra filter the
                         result follows HashProbe::evalFilterForNullAwareJoin (exec/HashProbe.cpp:1639-1700):
                         three-valued IN over the build rows that pass the filter */
  int32_t null_as_value; /* must equal the build side's */
  int32_t pad;
} vx355_join_probe_spec;

int vx355_join_probe_create(
    vx355_join_table* table,
    const vx355_join_probe_spec* spec,
    vx355_join_probe** out);

/* The join's extra filter (HashJoinNode::filter, evaluated by HashProbe::evalFilter,
 * exec/HashProbe.cpp:1713, on every (probe row, build row) candidate pair): a
 * conjunction of up to 4 comparisons between a probe column or a build-side
 * dependent column and a constant, a probe column or a build dependent. A null
 * operand fails the term. Integer-like and DATE columns compare as int64 (as
 * DOUBLE against REAL / DOUBLE), VARCHAR / VARBINARY (inline, <= 12 bytes) with
 * EQ / NE only. Semantics per join kind (HashProbe.cpp:1487-1711): INNER / RIGHT
 * keep the passing pairs; LEFT / FULL emit (row, null) when no pair of the row
 * passes; the semi kinds ask "does any pair pass", ANTI "does none"; probed flags
 * (right / full / right semi / right anti) are set for passing pairs only.
 * Call before the first add_input; the probe batch must stay alive until its
 * output is drained (the filter reads its columns at output time). */
typedef struct vx355_join_filter_term {
  int32_t left_side;  /* 0 = probe batch column, 1 = build dependent (index into dependent_cols) */
  int32_t left_col;
  int32_t cmp;        /* vx355_cmp */
  int32_t right_kind; /* 0 = constant, 1 = probe batch column, 2 = build dependent */
  int32_t right_col;
  int32_t const_kind; /* as vx355_filter_term */
  int32_t str_size;
  int32_t pad;
  int64_t i64;
  double f64;
  char str[16];
} vx355_join_filter_term;
int vx355_join_probe_set_filter(vx355_join_probe* h, const vx355_join_filter_term* terms, int32_t n_terms);
/* Operator fusion FilterProject -> HashProbe (ABI 7), the join-side twin of vx355_agg_set_fused_input:
 * the adapter replaces a FilterProject that only filters (FilterNode without computed projections:
 * every output column is an input column, exec/FilterProject.cpp:25-41) and the HashProbe behind it
 * with one operator. After this call the batches handed to vx355_join_probe_add_input are the
 * FilterProject's INPUT batches; a row failing the conjunction (vx355_filter_term, nulls fail) is
 * treated as a probe row that found nothing. mapping_out of get_output then numbers the rows of the
 * unfiltered batch - exactly the composition of FilterProject's selected indices with the probe's
 * mapping that the two separate operators produce (exec/OperatorUtils.cpp:393-422). TPC-H Q3 probes
 * lineitem WHERE l_shipdate > d and orders WHERE o_orderdate < d this way: no selection-bitmap pass,
 * no compaction, no index vector between the scan and the probe.
 * Only join kinds whose unmatched probe rows emit nothing (INNER, RIGHT, LEFT_SEMI_FILTER,
 * COUNTING_LEFT_SEMI_FILTER, RIGHT_SEMI_*, RIGHT_ANTI), not null aware; anything else is
 * VX355_EUNSUPPORTED and the FilterProject stays its own operator. Call before the first add_input;
 * up to 4 terms over columns of the probe batch. */
int vx355_join_probe_set_input_filter(vx355_join_probe* h, const vx355_filter_term* terms, int32_t n_terms);
/* QueryConfig::preferredOutputBatchBytes (core/QueryConfig.h:479), the second bound of
 * listJoinResults (exec/HashTable.cpp:2087-2153): get_output stops a page once the requested build
 * columns of its rows reach 'bytes' (at least one row per page). 0 = row bound only. */
int vx355_join_probe_set_output_batch_bytes(vx355_join_probe* h, int64_t bytes);
/* HashProbe::addInput (exec/HashProbe.cpp:796-900): prepareForJoinProbe
 * (HashTable.cpp:2680-2712) + joinProbe (:610-652) for the whole batch. */
int vx355_join_probe_add_input(vx355_join_probe* h, const vx355_batch* batch);
/* addInput for tables that no cache holds (ABI 8). The reference meets a table larger than the CPU's
 * caches by partitioning the BUILD over threads that each own a contiguous range of buckets
 * (parallelJoinBuild, exec/HashTable.cpp:1003-1203) and by keeping 64 probes in flight
 * (joinNormalizedKeyProbe, :697-725). On the GPU a probe of a table of hundreds of MB is bound by
 * the fabric's dependent-random-read rate, not by HBM bytes; the slot array is already cut into
 * contiguous slices by the top bits of the slot number, so this entry point regroups the PROBE side:
 * every column of 'batch' is moved into regrouped_values[c] (device buffers of num_rows x the
 * column's width, allocated by the caller) so that rows whose keys begin their walk in the same slice
 * (~4 MiB of slots: an XCD's L2) are adjacent, and the regrouped batch is probed slice by slice with each slice's
 * workgroups on one XCD, whose L2 then holds the slice. *regrouped = 1: the operator's input batch
 * IS the regrouped one - mapping_out of get_output numbers ITS rows (ascending, as always), so the
 * caller wraps regrouped_values, not the original columns; the buffers stay valid until the output
 * is drained. Inside a slice rows keep no particular order (an exchange does not define one either).
 * *regrouped = 0: the table or the batch does not qualify (array / generic mode, a table that
 * caches hold, a small batch, encodings other than FLAT, null bitmaps, a fused input filter) and the
 * batch was probed as vx355_join_probe_add_input does; regrouped_values are untouched.
 * All join kinds, extra filters and counting joins work on the regrouped batch as on any other.
 * Used by vx355_join_repartition for every received chunk. Environment: VX355_JOIN_REGROUP = 0 never,
 * 1 whenever eligible (tests), unset = tables >= 64 MiB and batches >= 4 M rows;
 * VX355_JOIN_SLICE_BYTES (default 4 MiB). */
int vx355_join_probe_add_input_regrouped(vx355_join_probe* h, const vx355_batch* batch, void* const* regrouped_values,
                                         int32_t* regrouped);
/* Asynchronous form (ABI 7), as for the aggregation and the build (see vx355_agg_add_input_async): the
 * upload of the batch, the probe kernels and the read-back of the output size run on the handle's
 * worker thread; the Driver thread returns at once (exec/Operator.h:285-299) and either polls
 * (isBlocked) or simply calls get_output, which waits. One batch in flight: the next add_input
 * follows the last get_output of this one. The batch's buffers stay valid until its output is
 * drained, as for the synchronous form. */
int vx355_join_probe_add_input_async(vx355_join_probe* h, const vx355_batch* batch, int64_t* ticket_out);
int vx355_join_probe_poll(vx355_join_probe* h, int64_t* submitted, int64_t* completed);
int vx355_join_probe_wait(vx355_join_probe* h);
/* Queued output page of the probe (ABI 7; the aggregation's vx355_agg_get_output_async, same callback type):
 * vx355_join_probe_get_output (build_side = 0) or vx355_join_probe_get_build_side_output (build_side = 1,
 * mapping_out unused) as a task behind the batch queued with vx355_join_probe_add_input_async. The descriptor
 * arrays are copied; mapping_out, build_rows_out and the column buffers must stay valid until the ticket
 * completes. vx355_join_probe_output_result: (num_rows, finished) or the failure, once, after
 * vx355_join_probe_poll reported completed >= ticket. */
int vx355_join_probe_get_output_async(vx355_join_probe* h, int32_t build_side, int32_t max_rows, int32_t* mapping_out,
                                      int32_t* build_rows_out, int32_t out_mem, const vx355_out_column* build_cols,
                                      const int32_t* build_col_ids, int32_t num_build_cols,
                                      vx355_output_done_fn done, void* done_arg, int64_t* ticket_out);
int vx355_join_probe_output_result(vx355_join_probe* h, int64_t ticket, int32_t* num_rows, int32_t* finished);
/* HashProbe::getOutput (:1154) -> listJoinResults (HashTable.cpp:2133-2350) +
 * fillOutput (HashProbe.cpp:968-991). Emits at most max_rows result rows in
 * ascending probe-row order, all matches of one probe row contiguous.
 * mapping_out[i] = probe row (the indices the shim wraps the probe columns
 * with, OperatorUtils.cpp:380-422); build_rows_out[i] = build row id or -1
 * for a miss (LEFT / ANTI); build_cols[j] receives dependent column
 * build_col_ids[j] gathered at build_rows_out (extractColumns,
 * HashProbe.cpp:82-118; null for misses). *finished = 1 when the current
 * input batch is drained. mapping_out/build_rows_out live in out_mem. */
int vx355_join_probe_get_output(
    vx355_join_probe* h,
    int32_t max_rows,
    int32_t* mapping_out,
    int32_t* build_rows_out,
    int32_t out_mem,
    vx355_out_column* build_cols,
    const int32_t* build_col_ids,
    int32_t num_build_cols,
    int32_t* n_out,
    int32_t* finished);
/* HashProbe::getBuildSideOutput (exec/HashProbe.cpp:993-1080): the build rows a
 * RIGHT / FULL / RIGHT_ANTI join has to emit (listNotProbedRows: no probe of ANY
 * vx355_join_probe of this table matched them, rows with null keys included),
 * the matched ones for RIGHT_SEMI_FILTER (listProbedRows), or every build row
 * for RIGHT_SEMI_PROJECT (listAllRows) whose 'match' column is requested as
 * build column id VX355_BUILD_COL_MATCH (BOOLEAN, never null: not null aware),
 * in ascending build-row order, max_rows at a time. Call it on ONE probe handle
 * (the reference's last prober, HashProbe.cpp:1189-1219) after every probe of
 * the table has consumed its input. LEFT_SEMI_PROJECT needs no such call: its
 * get_output lists every probe row once with build_rows_out = first match or -1
 * (the shim's 'match' column is build_rows_out >= 0); when null aware, -2 stands for a NULL
 * match: a null probe key against a build side that is not both empty and null-free, or no
 * match while the build side holds a null key (HashProbe::fillLeftSemiProjectMatchColumn,
 * exec/HashProbe.cpp:923-966). */
#define VX355_BUILD_COL_MATCH (-1)
int vx355_join_probe_get_build_side_output(
    vx355_join_probe* h,
    int32_t max_rows,
    int32_t* build_rows_out,
    int32_t out_mem,
    vx355_out_column* build_cols,
    const int32_t* build_col_ids,
    int32_t num_build_cols,
    int32_t* n_out,
    int32_t* finished);
void vx355_join_probe_destroy(vx355_join_probe* h);
void* vx355_join_probe_stream(vx355_join_probe* h);

/* ---- multi-GPU exchange (RCCL over xGMI) ------------------
 * The hot path shards without communication except for two steps: the exchange of a
 * repartitioned join (exec/PartitionedOutput.cpp + exec/Exchange.cpp in the reference,
 * experimental/ucx-exchange for its GPU backend) and the partial -> final merge of a
 * row-sharded aggregation (docs/develop/aggregations.rst:24-91). Both run inside the
 * library on communicators it owns; librccl is loaded at first use.
 *
 * A communicator is bound to one GPU and carries its own execution context.
 *  - one process per GPU: rank 0 calls vx355_comm_get_unique_id, the 128 bytes travel out
 *    of band (MPI, a file, torch.distributed ...), every rank calls vx355_comm_create on
 *    the thread bound to its GPU (vx355_set_device);
 *  - one process for all GPUs of the node: vx355_init each device, then
 *    vx355_comm_create_all; out[i] lives on devices[i]. Collective calls block until the
 *    peers arrive: drive each communicator from its own thread (one Driver per GPU). */
typedef struct vx355_comm vx355_comm;
#define VX355_COMM_ID_BYTES 128
int vx355_comm_get_unique_id(void* id_out /* VX355_COMM_ID_BYTES */);
int vx355_comm_create(const void* id, int32_t world, int32_t rank, vx355_comm** out);
int vx355_comm_create_all(int32_t num_devices, const int32_t* devices, vx355_comm** out /* num_devices */);
int vx355_comm_info(const vx355_comm* c, int32_t* world, int32_t* rank, int32_t* device);
void* vx355_comm_stream(vx355_comm* c);
void vx355_comm_destroy(vx355_comm* c);

/* Exchange of rows already grouped by destination rank (vx355_partition_scatter with
 * num_partitions = world): send_counts[p] rows go to rank p.
 * vx355_exchange_counts: recv_counts[s] = rows rank s sends to this rank (host arrays of
 *   'world' entries; one small all-gather).
 * vx355_exchange_columns: for every column (device buffers, widths[c] bytes per row) the
 *   slice of rank p goes straight to rank p and the slices of all ranks arrive in rank
 *   order: one ncclGroupStart / ncclSend + ncclRecv per peer and column / ncclGroupEnd, so
 *   every slice rides its own xGMI link. recv_cols[c] must hold sum(recv_counts) rows. */
int vx355_exchange_counts(vx355_comm* c, const int64_t* send_counts, int64_t* recv_counts);
int vx355_exchange_columns(
    vx355_comm* c,
    const void* const* send_cols,
    const int32_t* widths,
    int32_t num_cols,
    const int64_t* send_counts,
    const int64_t* recv_counts,
    void* const* recv_cols);
/* All-gather of bytes_per_rank bytes from every rank (device buffers; recv holds world
 * blocks in rank order): the partial results of a row-sharded aggregation meet on every
 * rank before its FINAL step. */
int vx355_all_gather(vx355_comm* c, const void* send, void* recv, size_t bytes_per_rank);
/* The same for blocks of different sizes: sizes[s] bytes arrive from rank s (sizes[rank] is
 * this rank's own block; a host array of 'world' entries, e.g. from vx355_exchange_counts with
 * every send count set to the own size), back to back in rank order. */
int vx355_all_gather_v(vx355_comm* c, const void* send, const int64_t* sizes, void* recv);

/* ---- the plan fragments around the exchange, orchestrated inside the library --------------
 * One plan edge "PartitionedOutput(keys) -> Exchange" of a repartitioned join
 * (exec/PartitionedOutput.cpp:59-133 with exec/HashPartitionFunction.cpp:76-118 on the sending
 * side, exec/Exchange.cpp on the receiving side), both halves on one handle per rank:
 *  - send (PartitionedOutput::addInput): VectorHasher::hash of the key columns, partition =
 *    the top log2(world) hash bits (world a power of two; disjoint from the bits the join tables
 *    index with, cf. checkHashBitsOverlap exec/HashTable.cpp:1853) or hash % world
 *    (HashPartitionFunction.cpp:112-115), rows grouped by destination (stable), slice sizes
 *    all-gathered, then every slice posted straight to its owner (grouped ncclSend / ncclRecv on
 *    a stream of the handle's own). The call returns while the slices are on the links; two
 *    sends may be in flight. Collective: every rank sends the same number of batches.
 *    Columns: FLAT, no nulls, fixed width (strings inline, <= 12 bytes); host or device memory.
 *  - receive (Exchange::getOutput): waits for the oldest send in flight and hands out the rows
 *    that landed on this rank as FLAT device columns, source ranks in rank order, input order
 *    kept inside a source. The buffers belong to the handle and stay valid until the next
 *    receive on it. */
typedef struct vx355_exchange vx355_exchange;
int vx355_exchange_create(vx355_comm* c, const int32_t* col_types, int32_t num_cols, const int32_t* key_cols,
                          int32_t num_keys, vx355_exchange** out);
int vx355_exchange_send(vx355_exchange* x, const vx355_batch* batch);
int vx355_exchange_receive(vx355_exchange* x, vx355_column* cols_out /* num_cols */, int64_t* rows_out);
/* The destination rank of every row of 'batch' as the edge computes it for num_destinations ranks
 * (0 = the communicator's size): VectorHasher::hash of the key columns fused with
 * HashPartitionFunction::partition, i.e. what vx355_exchange_send groups by. out: num_rows entries
 * in out_mem. (Inspection; lets a 1-GPU box check the N > 1 grouping against the CPU hash.) */
int vx355_exchange_destinations(vx355_exchange* x, const vx355_batch* batch, int32_t num_destinations,
                                uint32_t* out, int32_t out_mem);
void* vx355_exchange_stream(vx355_exchange* x);
void vx355_exchange_destroy(vx355_exchange* x);

/* The repartitioned join of BASELINE config 5 in one call (what velox_amd/dist.py did in
 * Python until ABI 3): build_rows / probe_rows are this rank's row-range shards (FLAT columns;
 * the specs' key_cols / dependent_cols index them). Build side: exchange, HashBuild::addInput,
 * noMoreInput -> *table_out (one reference, the caller releases it). Probe side: cut into
 * 'chunks' row ranges; the slices of chunk i are on the links while chunk i + 1 is hashed and
 * grouped and chunk i - 1 is probed. For every chunk the sink is called once, after
 * HashProbe::addInput of the rows that landed here: it drains 'probe' (vx355_join_probe_get_output)
 * before it returns; 'received' (device columns) is valid during the call. A non-zero return
 * aborts the join with that status. Collective: every rank calls it with the same 'chunks'.
 * 'received' is the operator's input batch - the rows that landed here in an order the library
 * chooses: since ABI 8 regrouped by slice of the join table when the table is beyond the caches
 * (vx355_join_probe_add_input_regrouped); mapping_out numbers its rows. With one rank (no links)
 * nothing is exchanged: the caller's rows go straight to the build and, regrouped, to the probe. */
typedef int (*vx355_join_chunk_sink)(void* arg, int32_t chunk, const vx355_batch* received, vx355_join_probe* probe);
int vx355_join_repartition(
    vx355_comm* c,
    const vx355_join_build_spec* build_spec,
    const vx355_batch* build_rows,
    const vx355_join_probe_spec* probe_spec,
    const vx355_batch* probe_rows,
    int32_t chunks,
    vx355_join_chunk_sink sink,
    void* sink_arg,
    vx355_join_table** table_out);

/* Partial -> final merge of a row-sharded aggregation (docs/develop/aggregations.rst:24-91):
 * 'partial' (PARTIAL / INTERMEDIATE step, after vx355_agg_no_more_input) is drained, its groups
 * leave as one PrestoPage per rank (the wire format of Velox's exchange, lossless timestamps),
 * every rank receives every page, and a new operator created from final_spec (FINAL or
 * INTERMEDIATE step; its column numbers refer to the partial operator's output layout) consumes
 * them in rank order and is returned after noMoreInput: the caller drains *final_out with
 * vx355_agg_get_output and destroys both operators. Collective. */
int vx355_agg_merge_partials(vx355_comm* c, vx355_agg* partial, const vx355_agg_spec* final_spec,
                             vx355_agg** final_out);

#ifdef __cplusplus
}
#endif
#endif /* VX355_H_ */
