// VectorHasher in distinct-value mode on the device (exec/VectorHasher.cpp:128-161,196-224,
// 906-921; VectorHasher.h:560-580): value ids are the 1-based insertion numbers of the distinct
// values of a key column, handed out in row order — the reference inserts values one row at a
// time into an F14 set and numbers them by its size. A vx355_value_dict is that set: an
// open-addressing table {value, id} in HBM that persists across batches.
//
// One batch:
//   1. every selected non-null row looks its value up; rows whose value is new are listed
//      (ascending row order);
//   2. the new (value, position in the list) pairs are radix sorted by value — stable, so the
//      first pair of every run is the value's first occurrence;
//   3. the runs are sorted by that first occurrence: rank r -> id = size + 1 + r, exactly the
//      number the row-at-a-time loop would have assigned; the new values are inserted;
//   4. the listed rows look their value up again; result[row] = id (multiplier 1) or
//      result[row] + multiplier * id.
// A value that makes the set reach 'range_size' entries is unmappable (VectorHasher.h:573-576):
// *all_mapped = 0 then, as in range mode.
#include "common.h"

#include <algorithm>

namespace vx {

void compactBits(const uint64_t* dValues, const uint64_t* dNulls, const uint64_t* dRows,
                 int64_t numRows, int32_t* dOut, DevBuf& scratch, int64_t* total);
void sortPairsU64U32(uint64_t* keys, uint32_t* vals, uint64_t* keysTmp, uint32_t* valsTmp,
                     size_t n, DevBuf& tmp, bool* resultInTmp, int endBit = 64);

namespace {

struct DictSlot {
  uint64_t value;  // sign-flipped int64 image (orders like the value)
  uint64_t id;     // 0 = free
};

__device__ inline uint64_t dictImage(const ColView& c, int64_t i, bool* ok) {
  int64_t v;
  if (c.kind == VX355_VARCHAR || c.kind == VX355_VARBINARY) {
    const StringView16 sv = loadView(c, i);
    if (sv.size > kStringAsRangeMaxSize) {
      *ok = false;
      return 0;
    }
    v = stringAsNumber(sv);
  } else {
    v = loadInt64(c, i);
  }
  return static_cast<uint64_t>(v) ^ 0x8000000000000000ULL;
}

__device__ inline uint64_t dictFind(const DictSlot* slots, uint64_t mask, uint64_t image) {
  uint64_t pos = twangMix64(image) & mask;
  for (uint64_t probes = 0; probes <= mask; ++probes) {
    const uint64_t id = slots[pos].id;
    if (id == 0) {
      return 0;
    }
    if (slots[pos].value == image) {
      return id;
    }
    pos = (pos + 1) & mask;
  }
  return 0;
}

struct DictLookupArgs {
  ColView col;
  const uint64_t* rows;      // selection or nullptr
  const DictSlot* slots;
  uint64_t mask;
  int64_t numRows;
  uint64_t multiplier;
  uint64_t* result;
  uint64_t* missBits;        // bit per row: selected, non-null, value not in the set
  uint32_t* unsupported;     // a string longer than 7 bytes
  int32_t writeNullZero;     // multiplier == 1: null rows get id 0
};

__global__ __launch_bounds__(256) void k_dict_lookup(DictLookupArgs a) {
  const int64_t numWords = (a.numRows + 63) >> 6;
  const int64_t waveStride = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t w = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; w < numWords;
       w += waveStride) {
    const int64_t row = (w << 6) + lane();
    bool miss = false;
    if (row < a.numRows && (!a.rows || bitAt(a.rows, row))) {
      if (colIsNull(a.col, row)) {
        if (a.writeNullZero) {
          a.result[row] = 0;
        }
      } else {
        bool ok = true;
        const uint64_t image = dictImage(a.col, colIndex(a.col, row), &ok);
        if (!ok) {
          *a.unsupported = 1;
        } else {
          const uint64_t id = dictFind(a.slots, a.mask, image);
          if (id == 0) {
            miss = true;
          } else {
            a.result[row] = a.multiplier == 1 ? id : a.result[row] + a.multiplier * id;
          }
        }
      }
    }
    const uint64_t word = ballot(miss);
    if (lane() == 0) {
      a.missBits[w] = word;
    }
  }
}

__global__ __launch_bounds__(256) void k_dict_images(ColView col, const int32_t* rows, int64_t n, uint64_t* keys,
                                                      uint32_t* vals) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    bool ok = true;
    keys[i] = dictImage(col, colIndex(col, rows[i]), &ok);
    vals[i] = static_cast<uint32_t>(i);
  }
}

// bit i = sorted[i] starts a run
__global__ __launch_bounds__(256) void k_dict_run_starts(const uint64_t* sorted, int64_t n, uint64_t* bits) {
  const int64_t numWords = (n + 63) >> 6;
  const int64_t waveStride = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t w = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; w < numWords;
       w += waveStride) {
    const int64_t i = (w << 6) + lane();
    const uint64_t word = ballot(i < n && (i == 0 || sorted[i] != sorted[i - 1]));
    if (lane() == 0) {
      bits[w] = word;
    }
  }
}

// per run: key = position of the value's first occurrence in the miss list, val = run number
__global__ __launch_bounds__(256) void k_dict_first_rows(const int32_t* runStart, int64_t numRuns,
                                                          const uint32_t* sortedVals, uint64_t* keys, uint32_t* vals) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; r < numRuns; r += stride) {
    keys[r] = sortedVals[runStart[r]];
    vals[r] = static_cast<uint32_t>(r);
  }
}

// runs in first-occurrence order: rank -> id; insert {value, id}
__global__ __launch_bounds__(256) void k_dict_insert(const uint32_t* runsByFirst, int64_t numRuns, const int32_t* runStart,
                                                      const uint64_t* sortedKeys, uint64_t firstId, DictSlot* slots,
                                                      uint64_t mask) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t rank = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; rank < numRuns; rank += stride) {
    const uint64_t image = sortedKeys[runStart[runsByFirst[rank]]];
    uint64_t pos = twangMix64(image) & mask;
    for (uint64_t probes = 0; probes <= mask; ++probes) {
      const unsigned long long old =
          atomicCAS(reinterpret_cast<unsigned long long*>(&slots[pos].id), 0ULL, firstId + static_cast<uint64_t>(rank));
      if (old == 0) {
        slots[pos].value = image;  // nobody looks the table up during this launch
        break;
      }
      pos = (pos + 1) & mask;
    }
  }
}

struct DictResolveArgs {
  ColView col;
  const int32_t* rows;
  int64_t n;
  const DictSlot* slots;
  uint64_t mask;
  uint64_t multiplier;
  uint64_t* result;
};

__global__ __launch_bounds__(256) void k_dict_resolve(DictResolveArgs a) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    const int64_t row = a.rows[i];
    bool ok = true;
    const uint64_t id = dictFind(a.slots, a.mask, dictImage(a.col, colIndex(a.col, row), &ok));
    a.result[row] = a.multiplier == 1 ? id : a.result[row] + a.multiplier * id;
  }
}

__global__ __launch_bounds__(256) void k_dict_clear_misses(const uint64_t* rows, const uint64_t* missBits, int64_t numWords,
                                                            uint64_t* rowsOut, int64_t numRows) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < numWords; w += stride) {
    uint64_t all = ~0ULL;
    if (w == numWords - 1 && (numRows & 63)) {
      all = (1ULL << (numRows & 63)) - 1;
    }
    rowsOut[w] = (rows ? rows[w] : all) & ~missBits[w];
  }
}

}  // namespace
}  // namespace vx

using namespace vx;

struct vx355_value_dict {
  vx::Runtime* ctx = nullptr;
  int32_t kind = 0;
  int64_t rangeSize = 0;
  int64_t size = 0;          // distinct values held (the unmappable ones included, like uniqueValues_)
  uint64_t capacity = 0;     // slots, power of two
  DevBuf slots, missBits, missRows, keys, vals, keysTmp, valsTmp, sortTmp, scratch, runBits, runStart, keys2, vals2,
      keys2Tmp, vals2Tmp, flag;
};

namespace {

void dictGrow(vx355_value_dict& d, uint64_t needValues) {
  auto& rt = Runtime::get();
  const uint64_t want = std::max<uint64_t>(1024, nextPow2(needValues * 2 + 2));
  if (want <= d.capacity) {
    return;
  }
  DevBuf fresh;
  fresh.ensure(static_cast<size_t>(want) * sizeof(DictSlot) + 64);
  HIP_OK(hipMemsetAsync(fresh.ptr(), 0, static_cast<size_t>(want) * sizeof(DictSlot), rt.stream));
  if (d.size > 0) {
    // re-insert the live entries (host round trip: the dictionary holds at most ~10^5 values)
    std::vector<DictSlot> old(d.capacity);
    copyOut(old.data(), VX355_MEM_HOST, d.slots.ptr(), static_cast<size_t>(d.capacity) * sizeof(DictSlot));
    std::vector<DictSlot> neu(want, DictSlot{0, 0});
    const uint64_t mask = want - 1;
    for (const auto& s : old) {
      if (s.id == 0) {
        continue;
      }
      uint64_t k = s.value;  // twang_mix64 on the host
      k = (~k) + (k << 21);
      k ^= k >> 24;
      k = k + (k << 3) + (k << 8);
      k ^= k >> 14;
      k = k + (k << 2) + (k << 4);
      k ^= k >> 28;
      k = k + (k << 31);
      uint64_t pos = k & mask;
      while (neu[pos].id != 0) {
        pos = (pos + 1) & mask;
      }
      neu[pos] = s;
    }
    copyIn(fresh.ptr(), neu.data(), VX355_MEM_HOST, static_cast<size_t>(want) * sizeof(DictSlot));
    rt.sync();
  }
  d.slots = std::move(fresh);
  d.capacity = want;
}

void dictRun(vx355_value_dict& d, const vx355_batch* batch, int32_t col, const uint64_t* rows, uint64_t multiplier,
             uint64_t* result, uint64_t* rowsOut, int32_t* allMapped, int32_t mem, bool lookupOnly) {
  auto& rt = Runtime::get();
  VX_CHECK_ARG(batch && result, "NULL argument");
  VX_CHECK_ARG(col >= 0 && col < batch->num_cols, "no such column");
  if (allMapped) {
    *allMapped = 1;
  }
  const int64_t n = batch->num_rows;
  if (n == 0) {
    return;
  }
  DeviceBatch db;
  db.load(batch, std::vector<int32_t>{col});
  const ColView cv = db.col(col);
  if (!(isIntLike(cv.kind) || isString(cv.kind)) || cv.kind != d.kind) {
    VX_THROW(VX355_EUNSUPPORTED, "value dictionary over a column of another type");
  }
  dictGrow(d, static_cast<uint64_t>(d.size) + 1);
  const size_t words = static_cast<size_t>(ceilDiv(n, 64));
  const bool host = mem == VX355_MEM_HOST;
  DevBuf dRows, dRes, dRowsOut;
  const uint64_t* devRows = rows;
  if (rows && host) {
    copyIn(dRows.ensure(words * 8 + 64), rows, VX355_MEM_HOST, words * 8);
    devRows = dRows.as<uint64_t>();
  }
  uint64_t* devRes = result;
  if (host) {
    copyIn(dRes.ensure(static_cast<size_t>(n) * 8 + 64), result, VX355_MEM_HOST, static_cast<size_t>(n) * 8);
    devRes = dRes.as<uint64_t>();
  }
  uint32_t* flag = static_cast<uint32_t*>(d.flag.ensure(64));
  HIP_OK(hipMemsetAsync(flag, 0, 4, rt.stream));
  DictLookupArgs la{};
  la.col = cv;
  la.rows = devRows;
  la.slots = d.slots.as<DictSlot>();
  la.mask = d.capacity - 1;
  la.numRows = n;
  la.multiplier = multiplier;
  la.result = devRes;
  la.missBits = static_cast<uint64_t*>(d.missBits.ensure(words * 8 + 64));
  la.unsupported = flag;
  la.writeNullZero = multiplier == 1 ? 1 : 0;
  VX_LAUNCH("k_dict_lookup", k_dict_lookup, streamGrid(static_cast<int64_t>(words) * 64, 256), 256, 0, la);
  uint32_t unsupported = 0;
  copyOut(&unsupported, VX355_MEM_HOST, flag, 4);
  if (unsupported) {
    VX_THROW(VX355_EUNSUPPORTED, "value ids over strings longer than 7 bytes");
  }
  if (lookupOnly) {
    // lookupValueIds (VectorHasher.cpp:550-565): unknown values are proven misses
    if (rowsOut) {
      uint64_t* devOut = host ? static_cast<uint64_t*>(dRowsOut.ensure(words * 8 + 64)) : rowsOut;
      VX_LAUNCH("k_dict_clear_misses", k_dict_clear_misses, streamGrid(static_cast<int64_t>(words), 256), 256, 0, devRows,
                la.missBits, static_cast<int64_t>(words), devOut, n);
      if (host) {
        copyOut(rowsOut, VX355_MEM_HOST, devOut, words * 8);
      }
    }
  } else {
    int32_t* missRows = static_cast<int32_t*>(d.missRows.ensure(static_cast<size_t>(n) * 4 + 64));
    int64_t numMiss = 0;
    compactBits(la.missBits, nullptr, nullptr, n, missRows, d.scratch, &numMiss);
    if (numMiss > 0) {
      const size_t m = static_cast<size_t>(numMiss);
      uint64_t* keys = static_cast<uint64_t*>(d.keys.ensure(m * 8 + 64));
      uint32_t* vals = static_cast<uint32_t*>(d.vals.ensure(m * 4 + 64));
      uint64_t* keysTmp = static_cast<uint64_t*>(d.keysTmp.ensure(m * 8 + 64));
      uint32_t* valsTmp = static_cast<uint32_t*>(d.valsTmp.ensure(m * 4 + 64));
      VX_LAUNCH("k_dict_images", k_dict_images, streamGrid(numMiss, 256), 256, 0, cv, missRows, numMiss, keys, vals);
      bool inTmp = false;
      sortPairsU64U32(keys, vals, keysTmp, valsTmp, m, d.sortTmp, &inTmp);
      const uint64_t* sortedKeys = inTmp ? keysTmp : keys;
      const uint32_t* sortedVals = inTmp ? valsTmp : vals;
      uint64_t* runBits = static_cast<uint64_t*>(d.runBits.ensure(static_cast<size_t>(ceilDiv(numMiss, 64)) * 8 + 64));
      VX_LAUNCH("k_dict_run_starts", k_dict_run_starts, streamGrid(numMiss, 256), 256, 0, sortedKeys, numMiss, runBits);
      int32_t* runStart = static_cast<int32_t*>(d.runStart.ensure(m * 4 + 64));
      int64_t numRuns = 0;
      compactBits(runBits, nullptr, nullptr, numMiss, runStart, d.scratch, &numRuns);
      const size_t r = static_cast<size_t>(numRuns);
      uint64_t* keys2 = static_cast<uint64_t*>(d.keys2.ensure(r * 8 + 64));
      uint32_t* vals2 = static_cast<uint32_t*>(d.vals2.ensure(r * 4 + 64));
      uint64_t* keys2Tmp = static_cast<uint64_t*>(d.keys2Tmp.ensure(r * 8 + 64));
      uint32_t* vals2Tmp = static_cast<uint32_t*>(d.vals2Tmp.ensure(r * 4 + 64));
      VX_LAUNCH("k_dict_first_rows", k_dict_first_rows, streamGrid(numRuns, 256), 256, 0, runStart, numRuns, sortedVals,
                keys2, vals2);
      bool inTmp2 = false;
      sortPairsU64U32(keys2, vals2, keys2Tmp, vals2Tmp, r, d.sortTmp, &inTmp2, 32);
      const uint32_t* runsByFirst = inTmp2 ? vals2Tmp : vals2;
      // The set keeps growing while this batch is analysed, also past range_size (the
      // reference inserts, then reports kUnmappable: VectorHasher.h:569-577).
      dictGrow(d, static_cast<uint64_t>(d.size + numRuns));
      VX_LAUNCH("k_dict_insert", k_dict_insert, streamGrid(numRuns, 256), 256, 0, runsByFirst, numRuns, runStart,
                sortedKeys, static_cast<uint64_t>(d.size) + 1, d.slots.as<DictSlot>(), d.capacity - 1);
      d.size += numRuns;
      DictResolveArgs ra{};
      ra.col = cv;
      ra.rows = missRows;
      ra.n = numMiss;
      ra.slots = d.slots.as<DictSlot>();
      ra.mask = d.capacity - 1;
      ra.multiplier = multiplier;
      ra.result = devRes;
      VX_LAUNCH("k_dict_resolve", k_dict_resolve, streamGrid(numMiss, 256), 256, 0, ra);
    }
    // unmappable = a NEW value whose insertion leaves the set at range_size entries or more
    // (a batch of known values maps, whatever the size: VectorHasher.h:569-577)
    if (allMapped && numMiss > 0 && d.size >= d.rangeSize) {
      *allMapped = 0;
    }
  }
  if (host) {
    copyOut(result, VX355_MEM_HOST, devRes, static_cast<size_t>(n) * 8);
  }
  rt.sync();
}

}  // namespace

extern "C" {

int vx355_value_dict_create(int32_t type_kind, int64_t range_size, vx355_value_dict** out) {
  VX_API_BEGIN
  VX_CHECK_ARG(out && range_size >= 1, "bad argument");
  if (!(isIntLike(type_kind) || isString(type_kind))) {
    VX_THROW(VX355_EUNSUPPORTED, "value ids need an integer-like or string key (VectorHasher.h:338-357)");
  }
  auto d = std::make_unique<vx355_value_dict>();
  d->kind = type_kind;
  d->rangeSize = range_size;
  d->ctx = Runtime::createContext();
  *out = d.release();
  VX_API_END
}

int vx355_value_dict_compute(vx355_value_dict* d, const vx355_batch* batch, int32_t col, const uint64_t* rows,
                             uint64_t multiplier, uint64_t* result, int32_t* all_mapped, int32_t mem) {
  VX_API_BEGIN_CTX(VX_CTX_OF(d))
  VX_CHECK_ARG(d, "NULL argument");
  dictRun(*d, batch, col, rows, multiplier, result, nullptr, all_mapped, mem, false);
  VX_API_END
}

int vx355_value_dict_lookup(vx355_value_dict* d, const vx355_batch* batch, int32_t col, const uint64_t* rows,
                            uint64_t multiplier, uint64_t* result, uint64_t* rows_out, int32_t mem) {
  VX_API_BEGIN_CTX(VX_CTX_OF(d))
  VX_CHECK_ARG(d, "NULL argument");
  dictRun(*d, batch, col, rows, multiplier, result, rows_out, nullptr, mem, true);
  VX_API_END
}

int64_t vx355_value_dict_size(const vx355_value_dict* d) { return d ? d->size : -1; }

void vx355_value_dict_destroy(vx355_value_dict* d) {
  if (!d) {
    return;
  }
  Runtime* ctx = d->ctx;
  try {
    vx::ContextScope scope(ctx);
    delete d;
  } catch (...) {
  }
  Runtime::destroyContext(ctx);
}

}  // extern "C"
