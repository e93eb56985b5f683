// Standalone kernels of the parity-test surface (include/vx355.h):
//   vx355_hash_columns   == VectorHasher::hash           (exec/VectorHasher.cpp:567-584)
//   vx355_value_ids      == computeValueIds/lookupValueIds in range mode (:354-360,:550-565)
//   vx355_filter_compact == processFilterResults, flat   (exec/OperatorUtils.cpp:231-257)
//   vx355_partition      == HashPartitionFunction::partition (exec/HashPartitionFunction.cpp:76-118)
// All four are streaming, HBM-bound kernels: one row per lane, coalesced
// reads of the flat value buffers, grid-stride over >= 8 blocks per CU.
#include "common.h"

namespace vx {

constexpr int kMaxKeys = 8;

struct HashArgs {
  ColView keys[kMaxKeys];
  int32_t numKeys;
  int32_t mixFirst;
  const uint64_t* rows;
  uint64_t* out;
  int64_t numRows;
};

__global__ __launch_bounds__(256) void k_hash_columns(HashArgs a) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; row < a.numRows;
       row += stride) {
    if (a.rows && !bitAt(a.rows, row)) {
      continue;
    }
    uint64_t h = 0;
    bool have = false;
    if (a.mixFirst) {
      h = a.out[row];
      have = true;
    }
    for (int k = 0; k < a.numKeys; ++k) {
      const ColView& c = a.keys[k];
      uint64_t hv = colIsNull(c, row) ? kNullHash : hashValueAt(c, colIndex(c, row));
      h = have ? hashMix(h, hv) : hv;
      have = true;
    }
    a.out[row] = h;
  }
}

struct ValueIdArgs {
  ColView keys[kMaxKeys];
  KeyRange ranges[kMaxKeys];
  int32_t numKeys;
  int32_t lookup;
  const uint64_t* rows;
  uint64_t* result;
  uint64_t* rowsOut;     // lookup: selection AND mapped, one word per 64 rows
  uint32_t* unmapped;    // !lookup: set to 1 when a selected value is out of range
  int64_t numRows;
};

// One wave covers 64 consecutive rows so the lookup variant can rebuild the
// selection word with a single ballot.
__global__ __launch_bounds__(256) void k_value_ids(ValueIdArgs a) {
  const int64_t numWords = (a.numRows + 63) >> 6;
  const int64_t waveStride = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  bool anyUnmapped = false;
  for (int64_t w = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; w < numWords;
       w += waveStride) {
    const int64_t row = (w << 6) + lane();
    bool selected = row < a.numRows && (!a.rows || bitAt(a.rows, row));
    bool mapped = selected;
    if (selected) {
      uint64_t acc = 0;
      bool loaded = false;
      bool write = false;
      for (int k = 0; k < a.numKeys; ++k) {
        const ColView& c = a.keys[k];
        const KeyRange& r = a.ranges[k];
        if (colIsNull(c, row)) {
          // exec/VectorHasher.cpp:204-210: a null writes 0 only for multiplier 1.
          if (r.multiplier == 1) {
            acc = 0;
            loaded = true;
            write = true;
          }
          continue;
        }
        int64_t value;
        bool mappable;
        uint64_t id = valueIdAt(c, colIndex(c, row), r, &value, &mappable);
        if (id == 0) {
          mapped = false;
          continue;
        }
        if (r.multiplier == 1) {
          acc = id;
          loaded = true;
        } else {
          if (!loaded) {
            acc = a.result[row];
            loaded = true;
          }
          acc += r.multiplier * id;
        }
        write = true;
      }
      if (write && (mapped || !a.lookup)) {
        a.result[row] = acc;
      }
      if (!mapped) {
        anyUnmapped = true;
      }
    }
    if (a.lookup && a.rowsOut) {
      uint64_t word = ballot(selected && mapped);
      if (lane() == 0) {
        a.rowsOut[w] = word;
      }
    }
  }
  if (!a.lookup && a.unmapped && anyUnmapped) {
    *a.unmapped = 1;
  }
}

// ---- filter compaction: selection bitmap -> ascending int32 row numbers ----
// Work unit = one wave = 64 selection words = 4096 rows.
__device__ inline uint64_t selectionWord(const uint64_t* values, const uint64_t* nulls,
                                         const uint64_t* rows, int64_t w, int64_t numWords,
                                         int64_t numRows) {
  if (w >= numWords) {
    return 0;
  }
  uint64_t s = values[w];
  if (nulls) {
    s &= nulls[w];
  }
  if (rows) {
    s &= rows[w];
  }
  if (w == numWords - 1 && (numRows & 63)) {
    s &= (1ULL << (numRows & 63)) - 1;
  }
  return s;
}

__global__ __launch_bounds__(256) void k_compact_count(const uint64_t* values, const uint64_t* nulls,
                                                        const uint64_t* rows, int64_t numRows,
                                                        int64_t numTiles, uint32_t* tileCounts) {
  const int64_t numWords = (numRows + 63) >> 6;
  const int64_t waveStride = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t t = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; t < numTiles;
       t += waveStride) {
    uint64_t s = selectionWord(values, nulls, rows, (t << 6) + lane(), numWords, numRows);
    int c = popc64(s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      c += __shfl_xor(c, off, kWave);
    }
    if (lane() == 0) {
      tileCounts[t] = static_cast<uint32_t>(c);
    }
  }
}

// Single-block exclusive scan of the tile counts (<= a few hundred thousand
// entries); total goes to the mailbox.
__global__ __launch_bounds__(1024) void k_scan_counts(const uint32_t* counts, int64_t n,
                                                       uint32_t* offsets, uint64_t* total) {
  __shared__ uint32_t partial[1024];
  const int t = threadIdx.x;
  const int64_t per = (n + blockDim.x - 1) / blockDim.x;
  const int64_t begin = t * per;
  const int64_t end = begin + per < n ? begin + per : n;
  uint32_t sum = 0;
  for (int64_t i = begin; i < end; ++i) {
    sum += counts[i];
  }
  partial[t] = sum;
  blockSync();
  // Hillis-Steele over 1024 partials.
  for (int off = 1; off < 1024; off <<= 1) {
    uint32_t v = t >= off ? partial[t - off] : 0;
    blockSync();
    partial[t] += v;
    blockSync();
  }
  uint32_t run = t == 0 ? 0 : partial[t - 1];
  for (int64_t i = begin; i < end; ++i) {
    offsets[i] = run;
    run += counts[i];
  }
  if (t == blockDim.x - 1) {
    *total = partial[1023];
  }
}

template <typename OffsetT>
__global__ __launch_bounds__(256) void k_compact_write(const uint64_t* values, const uint64_t* nulls,
                                                        const uint64_t* rows, int64_t numRows,
                                                        int64_t numTiles, const OffsetT* tileOffsets,
                                                        int32_t* out) {
  const int64_t numWords = (numRows + 63) >> 6;
  const int64_t waveStride = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t t = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; t < numTiles;
       t += waveStride) {
    uint64_t s = selectionWord(values, nulls, rows, (t << 6) + lane(), numWords, numRows);
    // Exclusive prefix of the popcounts across the wave.
    int c = popc64(s);
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int v = __shfl_up(incl, off, kWave);
      if (lane() >= off) {
        incl += v;
      }
    }
    uint32_t base = static_cast<uint32_t>(tileOffsets[t]) + static_cast<uint32_t>(incl - c);
    // Expand word by word: lane l owns bit l, so each store is a coalesced run.
    const int32_t rowBase = static_cast<int32_t>(t << 12);
    for (int j = 0; j < 64; ++j) {
      uint64_t wj = shfl64(s, j);
      if (wj == 0) {
        continue;
      }
      uint32_t bj = __shfl(base, j, kWave);
      if ((wj >> lane()) & 1) {
        int rank = popc64(wj & ((1ULL << lane()) - 1));
        out[bj + rank] = rowBase + (j << 6) + lane();
      }
    }
  }
}

struct PartitionArgs {
  const uint64_t* hashes;
  uint32_t* out;
  int64_t numRows;
  int32_t kind;
  uint32_t numPartitions;
  int32_t bitBegin;
  uint64_t mask;
};

__global__ __launch_bounds__(256) void k_partition(PartitionArgs a) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.numRows;
       i += stride) {
    uint64_t h = a.hashes[i];
    uint32_t p;
    switch (a.kind) {
      case VX355_PART_MODULO:
        p = static_cast<uint32_t>(h % a.numPartitions);
        break;
      case VX355_PART_BIT_RANGE:
        p = static_cast<uint32_t>((h >> a.bitBegin) & a.mask);
        break;
      case VX355_PART_LOCAL_MODULO:
        p = xxh32U32(reverseBitsPerByte(static_cast<uint32_t>(h)), 0) % a.numPartitions;
        break;
      default:
        p = static_cast<uint32_t>(
            (static_cast<uint64_t>(xxh32U32(reverseBitsPerByte(static_cast<uint32_t>(h)), 0)) >>
             a.bitBegin) &
            a.mask);
        break;
    }
    a.out[i] = p;
  }
}

// ---- exclusive scan u32 -> u64 for up to ~10^8 cells: block sums, a single-block
// scan of those, block-local scans plus base. offsets has n + 1 entries.
constexpr int kScanBlock = 1024;
constexpr int kScanPerThread = 8;
constexpr int kScanTile = kScanBlock * kScanPerThread;

__global__ __launch_bounds__(kScanBlock) void k_scan_block_sums(const uint32_t* in, int64_t n, uint64_t* sums) {
  __shared__ uint64_t partial[kScanBlock / 64];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanTile;
  uint64_t s = 0;
#pragma unroll
  for (int j = 0; j < kScanPerThread; ++j) {
    const int64_t i = base + j * kScanBlock + threadIdx.x;
    s += i < n ? in[i] : 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += shfl64(s, lane() ^ off);
  }
  if (lane() == 0) {
    partial[threadIdx.x >> 6] = s;
  }
  blockSync();
  if (threadIdx.x == 0) {
    uint64_t t = 0;
    for (int w = 0; w < kScanBlock / 64; ++w) {
      t += partial[w];
    }
    sums[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(1024) void k_scan_sums(uint64_t* sums, int64_t n, uint64_t* total) {
  __shared__ uint64_t partial[1024];
  const int t = threadIdx.x;
  const int64_t per = (n + blockDim.x - 1) / blockDim.x;
  const int64_t begin = t * per;
  const int64_t end = begin + per < n ? begin + per : n;
  uint64_t sum = 0;
  for (int64_t i = begin; i < end; ++i) {
    sum += sums[i];
  }
  partial[t] = sum;
  blockSync();
  for (int off = 1; off < 1024; off <<= 1) {
    uint64_t v = t >= off ? partial[t - off] : 0;
    blockSync();
    partial[t] += v;
    blockSync();
  }
  uint64_t run = t == 0 ? 0 : partial[t - 1];
  for (int64_t i = begin; i < end; ++i) {
    const uint64_t v = sums[i];
    sums[i] = run;
    run += v;
  }
  if (t == blockDim.x - 1) {
    *total = partial[1023];
  }
}

__global__ __launch_bounds__(kScanBlock) void k_scan_apply(const uint32_t* in, int64_t n, const uint64_t* blockBase,
                                                            uint64_t* out) {
  // Thread t owns kScanPerThread CONSECUTIVE cells so that a plain running sum
  // finishes the scan; the block-level part scans the per-thread totals.
  __shared__ uint64_t waveTotals[kScanBlock / 64];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanTile + static_cast<int64_t>(threadIdx.x) * kScanPerThread;
  uint32_t v[kScanPerThread];
  uint64_t mine = 0;
#pragma unroll
  for (int j = 0; j < kScanPerThread; ++j) {
    v[j] = base + j < n ? in[base + j] : 0;
    mine += v[j];
  }
  uint64_t incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint64_t o = shfl64(incl, lane() - off >= 0 ? lane() - off : lane());
    if (lane() >= off) {
      incl += o;
    }
  }
  if (lane() == 63) {
    waveTotals[threadIdx.x >> 6] = incl;
  }
  blockSync();
  uint64_t run = blockBase[blockIdx.x] + (incl - mine);
  for (int w = 0; w < (threadIdx.x >> 6); ++w) {
    run += waveTotals[w];
  }
#pragma unroll
  for (int j = 0; j < kScanPerThread; ++j) {
    if (base + j < n) {
      out[base + j] = run;
    }
    run += v[j];
  }
}

void scanU32ToU64(const uint32_t* in, int64_t n, uint64_t* out, DevBuf& scratch) {
  if (n <= 0) {
    return;
  }
  const int64_t blocks = ceilDiv(n, kScanTile);
  uint64_t* sums = static_cast<uint64_t*>(scratch.ensure(static_cast<size_t>(blocks) * 8 + 64));
  VX_LAUNCH("k_scan_block_sums", k_scan_block_sums, static_cast<int>(blocks), kScanBlock, 0, in, n, sums);
  VX_LAUNCH("k_scan_sums", k_scan_sums, 1, 1024, 0, sums, blocks, out + n);
  VX_LAUNCH("k_scan_apply", k_scan_apply, static_cast<int>(blocks), kScanBlock, 0, in, n, sums, out);
}

// Shared with the operators (agg.hip / join.hip).
void compactBits(const uint64_t* dValues, const uint64_t* dNulls, const uint64_t* dRows,
                 int64_t numRows, int32_t* dOut, DevBuf& scratch, int64_t* total) {
  auto& rt = Runtime::get();
  if (numRows == 0) {
    *total = 0;
    return;
  }
  const int64_t numTiles = ceilDiv(numRows, 4096);
  int grid = streamGrid(numTiles * 64, 256);
  if (numTiles > 16384) {
    // Large inputs: multi-block scan (the single-block one takes ~0.3 ms per 150 K tiles).
    const size_t countBytes = (static_cast<size_t>(numTiles) * 4 + 63) & ~static_cast<size_t>(63);
    char* base = static_cast<char*>(scratch.ensure(countBytes + static_cast<size_t>(numTiles + 1) * 8 + 64));
    uint32_t* counts = reinterpret_cast<uint32_t*>(base);
    uint64_t* offsets = reinterpret_cast<uint64_t*>(base + countBytes);
    DevBuf scanScratch;  // block-cache allocation: cheap
    VX_LAUNCH("k_compact_count", k_compact_count, grid, 256, 0, dValues, dNulls, dRows, numRows,
              numTiles, counts);
    scanU32ToU64(counts, numTiles, offsets, scanScratch);
    VX_LAUNCH("k_compact_write", k_compact_write<uint64_t>, grid, 256, 0, dValues, dNulls, dRows, numRows,
              numTiles, offsets, dOut);
    uint64_t sum = 0;
    copyOut(&sum, VX355_MEM_HOST, offsets + numTiles, 8);
    *total = static_cast<int64_t>(sum);
    return;
  }
  uint32_t* counts = static_cast<uint32_t*>(scratch.ensure(static_cast<size_t>(numTiles) * 8 + 64));
  uint32_t* offsets = counts + numTiles;
  VX_LAUNCH("k_compact_count", k_compact_count, grid, 256, 0, dValues, dNulls, dRows, numRows,
            numTiles, counts);
  VX_LAUNCH("k_scan_counts", k_scan_counts, 1, 1024, 0, counts, numTiles, offsets, rt.mail.dev);
  VX_LAUNCH("k_compact_write", k_compact_write<uint32_t>, grid, 256, 0, dValues, dNulls, dRows, numRows,
            numTiles, offsets, dOut);
  rt.sync();
  *total = static_cast<int64_t>(rt.mail.host[0]);
}


// wrap-of-a-wrap (exec/OperatorUtils.cpp:393-422 wrapChild over an already wrapped vector, what
// HashProbe::fillOutput does with FilterProject's output): out[i] = inner[outer[i]].
__global__ __launch_bounds__(256) void k_compose_indices(const int32_t* inner, int32_t innerSize, const int32_t* outer,
                                                         int64_t n, int32_t* out, uint32_t* bad) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  bool any = false;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int32_t at = __builtin_nontemporal_load(outer + i);
    const bool ok = at >= 0 && at < innerSize;
    any = any || !ok;
    out[i] = ok ? inner[at] : 0;
  }
  if (any) {
    *bad = 1;
  }
}

}  // namespace vx

using namespace vx;

extern "C" {

int vx355_hash_columns(const vx355_batch* batch, const int32_t* key_cols, int32_t n_keys,
                       const uint64_t* rows, int32_t mix_first, uint64_t* out, int32_t out_mem) {
  VX_API_BEGIN
  auto& rt = Runtime::get();
  rt.requireInit();
  VX_CHECK_ARG(batch && key_cols && out, "NULL argument");
  VX_CHECK_ARG(n_keys >= 1 && n_keys <= kMaxKeys, "1..8 key columns supported");
  const int64_t n = batch->num_rows;
  if (n == 0) {
    return VX355_OK;
  }
  DeviceBatch db;
  db.load(batch, std::vector<int32_t>(key_cols, key_cols + n_keys));
  HashArgs a{};
  for (int k = 0; k < n_keys; ++k) {
    a.keys[k] = db.col(key_cols[k]);
  }
  a.numKeys = n_keys;
  a.mixFirst = mix_first;
  a.numRows = n;
  DevBuf dRows, dOut;
  const size_t words = static_cast<size_t>(ceilDiv(n, 64));
  if (rows && out_mem == VX355_MEM_HOST) {
    copyIn(dRows.ensure(words * 8), rows, VX355_MEM_HOST, words * 8);
    a.rows = dRows.as<uint64_t>();
  } else {
    a.rows = rows;
  }
  if (out_mem == VX355_MEM_HOST) {
    // Unselected slots keep the caller's bytes: seed the scratch with them.
    copyIn(dOut.ensure(static_cast<size_t>(n) * 8), out, VX355_MEM_HOST, static_cast<size_t>(n) * 8);
    a.out = dOut.as<uint64_t>();
  } else {
    a.out = out;
  }
  VX_LAUNCH("k_hash_columns", k_hash_columns, streamGrid(n, 256), 256, 0, a);
  if (out_mem == VX355_MEM_HOST) {
    copyOut(out, VX355_MEM_HOST, a.out, static_cast<size_t>(n) * 8);
  }
  rt.sync();
  VX_API_END
}

int vx355_value_ids(const vx355_batch* batch, const int32_t* key_cols,
                    const vx355_value_id_spec* specs, int32_t n_keys, const uint64_t* rows,
                    int32_t lookup, uint64_t* result, uint64_t* rows_out, int32_t* all_mapped,
                    int32_t out_mem) {
  VX_API_BEGIN
  auto& rt = Runtime::get();
  rt.requireInit();
  VX_CHECK_ARG(batch && key_cols && specs && result, "NULL argument");
  VX_CHECK_ARG(n_keys >= 1 && n_keys <= kMaxKeys, "1..8 key columns supported");
  const int64_t n = batch->num_rows;
  if (all_mapped) {
    *all_mapped = 1;
  }
  if (n == 0) {
    return VX355_OK;
  }
  DeviceBatch db;
  db.load(batch, std::vector<int32_t>(key_cols, key_cols + n_keys));
  ValueIdArgs a{};
  for (int k = 0; k < n_keys; ++k) {
    a.keys[k] = db.col(key_cols[k]);
    const int32_t kind = a.keys[k].kind;
    if (!(isIntLike(kind) || isString(kind))) {
      VX_THROW(VX355_EUNSUPPORTED, "value ids need an integer-like or string key");
    }
    a.ranges[k].min = specs[k].min;
    a.ranges[k].max = specs[k].max;
    a.ranges[k].multiplier = specs[k].multiplier;
  }
  a.numKeys = n_keys;
  a.lookup = lookup;
  a.numRows = n;
  const size_t words = static_cast<size_t>(ceilDiv(n, 64));
  const bool host = out_mem == VX355_MEM_HOST;
  DevBuf dRows, dRes, dRowsOut;
  if (rows && host) {
    copyIn(dRows.ensure(words * 8), rows, VX355_MEM_HOST, words * 8);
    a.rows = dRows.as<uint64_t>();
  } else {
    a.rows = rows;
  }
  if (host) {
    copyIn(dRes.ensure(static_cast<size_t>(n) * 8), result, VX355_MEM_HOST,
           static_cast<size_t>(n) * 8);
    a.result = dRes.as<uint64_t>();
  } else {
    a.result = result;
  }
  if (lookup && rows_out) {
    a.rowsOut = host ? static_cast<uint64_t*>(dRowsOut.ensure(words * 8)) : rows_out;
  }
  uint32_t* flag = reinterpret_cast<uint32_t*>(rt.mail.dev);
  rt.mail.host[0] = 0;
  a.unmapped = flag;
  VX_LAUNCH("k_value_ids", k_value_ids, streamGrid(static_cast<int64_t>(words) * 64, 256), 256, 0, a);
  if (host) {
    copyOut(result, VX355_MEM_HOST, a.result, static_cast<size_t>(n) * 8);
    if (a.rowsOut) {
      copyOut(rows_out, VX355_MEM_HOST, a.rowsOut, words * 8);
    }
  }
  rt.sync();
  if (all_mapped && !lookup) {
    *all_mapped = (rt.mail.host[0] & 0xffffffffULL) ? 0 : 1;
  }
  VX_API_END
}

int vx355_filter_compact(const uint64_t* values, const uint64_t* nulls, const uint64_t* rows,
                         int32_t num_rows, int32_t* idx_out, int32_t* n_out, int32_t mem) {
  VX_API_BEGIN
  auto& rt = Runtime::get();
  rt.requireInit();
  VX_CHECK_ARG(n_out != nullptr, "n_out is NULL");
  VX_CHECK_ARG(num_rows >= 0, "negative num_rows");
  *n_out = 0;
  if (num_rows == 0) {
    return VX355_OK;
  }
  VX_CHECK_ARG(values && idx_out, "NULL argument");
  const size_t words = static_cast<size_t>(ceilDiv(num_rows, 64));
  const bool host = mem == VX355_MEM_HOST;
  DevBuf dIn, dOut, scratch;
  const uint64_t *dv = values, *dn = nulls, *dr = rows;
  if (host) {
    uint64_t* base = static_cast<uint64_t*>(dIn.ensure(words * 8 * 3));
    copyIn(base, values, VX355_MEM_HOST, words * 8);
    dv = base;
    if (nulls) {
      copyIn(base + words, nulls, VX355_MEM_HOST, words * 8);
      dn = base + words;
    }
    if (rows) {
      copyIn(base + 2 * words, rows, VX355_MEM_HOST, words * 8);
      dr = base + 2 * words;
    }
  }
  int32_t* dIdx = host ? static_cast<int32_t*>(dOut.ensure(static_cast<size_t>(num_rows) * 4)) : idx_out;
  int64_t total = 0;
  compactBits(dv, dn, dr, num_rows, dIdx, scratch, &total);
  if (host) {
    copyOut(idx_out, VX355_MEM_HOST, dIdx, static_cast<size_t>(total) * 4);
  }
  *n_out = static_cast<int32_t>(total);
  VX_API_END
}

int vx355_compose_indices(const int32_t* inner, int32_t inner_size, const int32_t* outer, int32_t num_rows,
                          int32_t* out, int32_t mem) {
  VX_API_BEGIN
  auto& rt = Runtime::get();
  rt.requireInit();
  VX_CHECK_ARG(num_rows >= 0 && inner_size >= 0, "negative size");
  if (num_rows == 0) {
    return VX355_OK;
  }
  VX_CHECK_ARG(inner && outer && out, "NULL argument");
  const bool host = mem == VX355_MEM_HOST;
  DevBuf dIn, dOut;
  const int32_t *di = inner, *dp = outer;
  int32_t* dout = out;
  if (host) {
    int32_t* base = static_cast<int32_t*>(dIn.ensure((static_cast<size_t>(inner_size) + num_rows) * 4 + 64));
    copyIn(base, inner, VX355_MEM_HOST, static_cast<size_t>(inner_size) * 4);
    copyIn(base + inner_size, outer, VX355_MEM_HOST, static_cast<size_t>(num_rows) * 4);
    di = base;
    dp = base + inner_size;
    dout = static_cast<int32_t*>(dOut.ensure(static_cast<size_t>(num_rows) * 4 + 64));
  }
  rt.mail.host[0] = 0;
  VX_LAUNCH("k_compose_indices", k_compose_indices, streamGrid(num_rows, 256, 4), 256, 0, di, inner_size, dp,
            static_cast<int64_t>(num_rows), dout, reinterpret_cast<uint32_t*>(rt.mail.dev));
  if (host) {
    copyOut(out, VX355_MEM_HOST, dout, static_cast<size_t>(num_rows) * 4);
  } else {
    rt.sync();
  }
  if (rt.mail.host[0] & 0xffffffffULL) {
    VX_THROW(VX355_EINVAL, "vx355_compose_indices: an outer index lies outside the inner index vector");
  }
  VX_API_END
}

int vx355_partition(const uint64_t* hashes, int32_t num_rows, int32_t kind, int32_t num_partitions,
                    int32_t bit_begin, int32_t bit_end, uint32_t* partitions_out, int32_t mem) {
  VX_API_BEGIN
  auto& rt = Runtime::get();
  rt.requireInit();
  VX_CHECK_ARG(num_rows >= 0, "negative num_rows");
  VX_CHECK_ARG(kind >= VX355_PART_MODULO && kind <= VX355_PART_LOCAL_BIT_RANGE, "bad partition kind");
  const bool bits = kind == VX355_PART_BIT_RANGE || kind == VX355_PART_LOCAL_BIT_RANGE;
  if (bits) {
    VX_CHECK_ARG(bit_begin >= 0 && bit_end > bit_begin && bit_end <= 64, "bad bit range");
  } else {
    VX_CHECK_ARG(num_partitions > 0, "num_partitions must be positive");
  }
  if (num_rows == 0) {
    return VX355_OK;
  }
  VX_CHECK_ARG(hashes && partitions_out, "NULL argument");
  const bool host = mem == VX355_MEM_HOST;
  DevBuf dIn, dOut;
  PartitionArgs a{};
  a.numRows = num_rows;
  a.kind = kind;
  a.numPartitions = static_cast<uint32_t>(num_partitions);
  a.bitBegin = bit_begin;
  const int width = bit_end - bit_begin;
  a.mask = width >= 64 ? ~0ULL : ((1ULL << width) - 1);
  if (host) {
    copyIn(dIn.ensure(static_cast<size_t>(num_rows) * 8), hashes, VX355_MEM_HOST,
           static_cast<size_t>(num_rows) * 8);
    a.hashes = dIn.as<uint64_t>();
    a.out = static_cast<uint32_t*>(dOut.ensure(static_cast<size_t>(num_rows) * 4));
  } else {
    a.hashes = hashes;
    a.out = partitions_out;
  }
  VX_LAUNCH("k_partition", k_partition, streamGrid(num_rows, 256), 256, 0, a);
  if (host) {
    copyOut(partitions_out, VX355_MEM_HOST, a.out, static_cast<size_t>(num_rows) * 4);
  }
  rt.sync();
  VX_API_END
}

}  // extern "C"
