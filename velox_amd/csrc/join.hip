// HashBuild / HashProbe on the MI355X (replaces exec/HashBuild.cpp:442-993,
// exec/HashTable.cpp:610-725,1394-1538,1989-2350 and exec/HashProbe.cpp
// :796-991 for the join kinds listed in include/vx355.h).
//
// Design:
//  * The build side is kept columnar in HBM: per key the int64 image the
//    reference's VectorHasher would normalise (value, stringAsNumber, bool),
//    per dependent column the raw values plus one validity byte per row. Rows
//    with a null key are dropped while appending (HashBuild.cpp:475-494) with
//    an order-preserving ballot compaction, so build row ids are dense and
//    follow input order.
//  * finish() merges peer builds, takes exact key ranges (reserve 0 for join
//    builds, HashTable.h:1213-1215) and picks
//      - array mode: head[normalized key] = build row (u32). The reference
//        caps this at 2 M entries for CPU caches (HashTable.h:146); with 288 GB
//        of HBM the cap is a multiple of the build size instead, or
//      - normalized-key mode: open addressing over 16-byte slots
//        {u64 key, u32 head row, u32 pad}, load factor <= 0.7, linear probing
//        from twang_mix64(key), key claimed by one CAS.
//    Rows with equal keys are chained through next[row] (HashTable.cpp
//    :1394-1418 pushNext / arrayPushRow).
//  * Probe: one lane per probe row computes the normalized key with
//    lookupValueIds semantics (out of range = proven miss, VectorHasher.cpp
//    :550-565), reads one head word (array) or walks slots (hash) and stores
//    hits[row]; duplicate tables also store the chain length. getOutput
//    (HashTable::listJoinResults, HashTable.cpp:2133-2350) turns per-row
//    counts into offsets with a two-level scan and emits (probe row, build row)
//    pairs in ascending probe-row order, resumable at any max_rows.
#include "common.h"
#include "expr_device.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>

namespace vx {

std::mutex gNoTableMutex;

void compactBits(const uint64_t* dValues, const uint64_t* dNulls, const uint64_t* dRows,
                 int64_t numRows, int32_t* dOut, DevBuf& scratch, int64_t* total);
void sortKeysU64(const uint64_t* in, uint64_t* out, size_t n, DevBuf& tmp, int beginBit = 0, int endBit = 64);  // radix_sort.hip
void makeTermArgs(const DeviceBatch& db, const vx355_filter_term* terms, int32_t n, TermArg* out);  // exprs.hip

namespace {

constexpr int kMaxKeys = 8;
constexpr int kMaxDeps = 16;
constexpr uint32_t kNoRow32 = 0xffffffffu;
constexpr uint64_t kEmptyKey = ~0ULL;

enum JoinMode : int32_t { JMODE_HASH = 0, JMODE_ARRAY = 1, JMODE_NORMALIZED = 2 };

struct BuildCounters {
  uint32_t unmappable;
  uint32_t longString;
  uint32_t nullKeyRows;
  uint32_t duplicates;
  uint32_t numDistinct;
  uint32_t tableFull;
  uint32_t depNulls;   // bit d: dependent column d holds a null in some build row
  uint32_t pad;
  int64_t keyMin[kMaxKeys];
  int64_t keyMax[kMaxKeys];
};

// ---- build: append ---------------------------------------------------------------
struct ValidArgs {
  ColView keys[kMaxKeys];
  int32_t numKeys;
  int64_t numRows;
  uint64_t* validWords;
  BuildCounters* counters;
};

// One word of "all keys non-null" per 64 rows.
__global__ __launch_bounds__(256) void k_key_valid(ValidArgs a) {
  const int64_t numWords = (a.numRows + 63) >> 6;
  const int64_t waveStride = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  uint32_t nulls = 0;
  for (int64_t w = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; w < numWords;
       w += waveStride) {
    const int64_t row = (w << 6) + lane();
    bool ok = row < a.numRows;
    if (ok) {
      for (int k = 0; k < a.numKeys; ++k) {
        if (colIsNull(a.keys[k], row)) {
          ok = false;
          break;
        }
      }
      if (!ok) {
        ++nulls;
      }
    }
    uint64_t m = ballot(ok);
    if (lane() == 0) {
      a.validWords[w] = m;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    nulls += __shfl_xor(nulls, off, kWave);
  }
  if (lane() == 0 && nulls) {
    atomicAdd(&a.counters->nullKeyRows, nulls);
  }
}

struct AppendArgs {
  ColView keys[kMaxKeys];
  ColView deps[kMaxDeps];
  uint64_t* keyOut[kMaxKeys];   // key images: 1 word, 2 for strings / timestamps
  int32_t keyWords[kMaxKeys];
  char* depOut[kMaxDeps];
  uint8_t* depValid[kMaxDeps];
  int32_t depWidth[kMaxDeps];
  int32_t numKeys;
  int32_t numDeps;
  const int32_t* rows;  // selected input rows, ascending; nullptr = all rows
  int64_t count;
  int64_t base;         // first build row id of this batch
  BuildCounters* counters;
  // right / full joins keep rows with null keys (they reach the output as
  // unmatched rows, HashBuild.cpp:475-494) but never enter the table:
  const uint64_t* keyValidWords;  // bit per input row: all keys non-null
  uint8_t* keyNullOut;            // per build row
  // HashJoinNode::isNullAsValue (IS NOT DISTINCT FROM keys, set operations): a null key is a
  // value (id 0 / kNullHash), the row is inserted, keyNullOut[row] = bit k for a null key k
  int32_t nullAsValue;
  // strings longer than 12 bytes (keys and payloads) are copied into the build side's arena
  char* arenaBase;
  unsigned long long* arenaCursor;
  uint64_t arenaCap;
};

// Copies a non-inline string into the arena; returns the new pointer (0: arena exhausted).
__device__ inline uint64_t arenaCopy(const AppendArgs& a, uint64_t srcPtr, uint32_t size) {
  const uint32_t padded = (size + 7) & ~7u;
  const unsigned long long at = atomicAdd(a.arenaCursor, static_cast<unsigned long long>(padded));
  if (!a.arenaBase || at + padded > a.arenaCap) {
    return 0;
  }
  const uint8_t* src = reinterpret_cast<const uint8_t*>(srcPtr);
  uint8_t* dst = reinterpret_cast<uint8_t*>(a.arenaBase + at);
  for (uint32_t i = 0; i < size; ++i) {
    dst[i] = src[i];
  }
  return reinterpret_cast<uint64_t>(dst);
}

// Arena bytes the selected rows of a batch need.
__global__ __launch_bounds__(256) void k_build_long_bytes(AppendArgs a, unsigned long long* total) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  unsigned long long mine = 0;
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < a.count; p += stride) {
    const int64_t row = a.rows ? a.rows[p] : p;
    for (int k = 0; k < a.numKeys + a.numDeps; ++k) {
      const ColView& c = k < a.numKeys ? a.keys[k] : a.deps[k - a.numKeys];
      if ((c.kind == VX355_VARCHAR || c.kind == VX355_VARBINARY) && !colIsNull(c, row)) {
        const uint32_t size = static_cast<const uint4*>(c.values)[colIndex(c, row)].x;
        if (size > 12) {
          mine += (size + 7) & ~7u;
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mine += shfl64(mine, lane() ^ off);
  }
  if (lane() == 0 && mine) {
    atomicAdd(total, mine);
  }
}

__global__ __launch_bounds__(256) void k_build_append(AppendArgs a) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  int64_t mn[kMaxKeys], mx[kMaxKeys];
  for (int k = 0; k < kMaxKeys; ++k) {
    mn[k] = INT64_MAX;
    mx[k] = INT64_MIN;
  }
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < a.count;
       p += stride) {
    const int64_t row = a.rows ? a.rows[p] : p;
    const bool keyOk = !a.keyValidWords || bitAt(a.keyValidWords, row);
    uint8_t nullMask = keyOk ? 0 : 1;
    if (a.nullAsValue) {
      nullMask = 0;
      for (int k = 0; k < a.numKeys; ++k) {
        nullMask |= colIsNull(a.keys[k], row) ? static_cast<uint8_t>(1u << k) : 0;
      }
    }
    if (a.keyNullOut) {
      a.keyNullOut[a.base + p] = nullMask;
    }
    for (int k = 0; k < a.numKeys; ++k) {
      const ColView& c = a.keys[k];
      if (a.nullAsValue ? ((nullMask >> k) & 1) != 0 : !keyOk) {
        a.keyOut[k][(a.base + p) * a.keyWords[k]] = 0;
        if (a.keyWords[k] == 2) {
          a.keyOut[k][(a.base + p) * 2 + 1] = 0;
        }
        continue;
      }
      const int64_t i = colIndex(c, row);
      uint64_t w0, w1;
      bool inlineOk = true;
      keyImage(c, i, &w0, &w1, &inlineOk);
      if (!inlineOk) {
        w1 = arenaCopy(a, w1, static_cast<uint32_t>(w0));
        if (w1 == 0) {
          a.counters->longString = 1;  // arena exhausted (host sized it from k_build_long_bytes)
        }
      }
      a.keyOut[k][(a.base + p) * a.keyWords[k]] = w0;
      if (a.keyWords[k] == 2) {
        a.keyOut[k][(a.base + p) * 2 + 1] = w1;
      }
      // Range statistics for the normalized-key decision (VectorHasher::analyze).
      if (c.kind == VX355_REAL || c.kind == VX355_DOUBLE || c.kind == VX355_TIMESTAMP) {
        a.counters->unmappable = 1;
        continue;
      }
      int64_t v;
      if (c.kind == VX355_VARCHAR || c.kind == VX355_VARBINARY) {
        KeyRange all;
        all.min = INT64_MIN;
        all.max = INT64_MAX;
        bool mappable;
        valueIdAt(c, i, all, &v, &mappable);
        if (!mappable) {
          a.counters->unmappable = 1;
          continue;
        }
      } else {
        v = static_cast<int64_t>(w0);  // integer kinds: the image is the widened value
      }
      mn[k] = v < mn[k] ? v : mn[k];
      mx[k] = v > mx[k] ? v : mx[k];
    }
    for (int d = 0; d < a.numDeps; ++d) {
      const ColView& c = a.deps[d];
      const bool valid = !colIsNull(c, row);
      a.depValid[d][a.base + p] = valid ? 1 : 0;
      if (!valid && !((__hip_atomic_load(&a.counters->depNulls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> d) & 1)) {
        atomicOr(&a.counters->depNulls, 1u << d);
      }
      const int w = a.depWidth[d];
      char* dst = a.depOut[d] + (a.base + p) * (w == 0 ? 1 : w);
      const int64_t i = valid ? colIndex(c, row) : 0;
      // fixed-width payloads move as whole words (both sides are naturally aligned)
      if (w == 8) {
        *reinterpret_cast<uint64_t*>(dst) = valid ? static_cast<const uint64_t*>(c.values)[i] : 0;
      } else if (w == 4) {
        *reinterpret_cast<uint32_t*>(dst) = valid ? static_cast<const uint32_t*>(c.values)[i] : 0;
      } else if (w == 2) {
        *reinterpret_cast<uint16_t*>(dst) = valid ? static_cast<const uint16_t*>(c.values)[i] : 0;
      } else if (w == 1) {
        dst[0] = valid ? static_cast<const char*>(c.values)[i] : 0;
      } else if (w == 0) {
        dst[0] = (valid && bitAt(static_cast<const uint64_t*>(c.values), i)) ? 1 : 0;
      } else {  // 16: StringView or Timestamp
        uint4 raw = valid ? static_cast<const uint4*>(c.values)[i] : make_uint4(0, 0, 0, 0);
        if (valid && (c.kind == VX355_VARCHAR || c.kind == VX355_VARBINARY) && raw.x > 12) {
          const uint64_t p2 = arenaCopy(a, (static_cast<uint64_t>(raw.w) << 32) | raw.z, raw.x);
          if (p2 == 0) {
            a.counters->longString = 1;
          }
          raw.z = static_cast<uint32_t>(p2);
          raw.w = static_cast<uint32_t>(p2 >> 32);
        }
        *reinterpret_cast<uint4*>(dst) = raw;
      }
    }
  }
  for (int k = 0; k < a.numKeys; ++k) {
    if (mn[k] <= mx[k]) {
      atomicMin(reinterpret_cast<long long*>(&a.counters->keyMin[k]), static_cast<long long>(mn[k]));
      atomicMax(reinterpret_cast<long long*>(&a.counters->keyMax[k]), static_cast<long long>(mx[k]));
    }
  }
}

// The common build side - INTEGER / BIGINT keys and 4- or 8-byte dependents, flat or dictionary-wrapped,
// none with a null bitmap, every row selected - without k_build_append's per-row interpretation: FOUR rows per
// lane, every load of an iteration issued before the first store. The interpreting kernel keeps one
// or two loads per lane in flight (a row after the other, each behind its kind switches) and moved
// TPC-H Q3's 14.6 M-row build side at 0.9 TB/s: 0.26 of that join's 1.67 ms (round 6).
constexpr int kAppendRows = 4;
template <int NK, int ND>
__global__ __launch_bounds__(256) void k_build_append_flat(AppendArgs a, uint32_t key32, uint32_t dep32, uint32_t keyDict,
                                                          uint32_t depDict) {
  const int64_t tile = 256LL * kAppendRows;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * tile;
  int64_t mn[NK], mx[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    mn[k] = INT64_MAX;
    mx[k] = INT64_MIN;
  }
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * tile + threadIdx.x; base < a.count; base += stride) {
    int64_t kv[kAppendRows][NK];
    uint64_t dv[kAppendRows][ND > 0 ? ND : 1];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
#pragma unroll
      for (int u = 0; u < kAppendRows; ++u) {
        const int64_t row = base + u * 256LL;
        int64_t at = row < a.count ? row : a.count - 1;  // clamped: no load under a lane predicate
        if ((keyDict >> k) & 1) {
          at = a.keys[k].indices[at];  // dictionary-wrapped (what a FilterProject or a probe hands on)
        }
        kv[u][k] = ((key32 >> k) & 1) ? static_cast<int64_t>(static_cast<const int32_t*>(a.keys[k].values)[at])
                                      : static_cast<const int64_t*>(a.keys[k].values)[at];
      }
    }
#pragma unroll
    for (int d = 0; d < ND; ++d) {
#pragma unroll
      for (int u = 0; u < kAppendRows; ++u) {
        const int64_t row = base + u * 256LL;
        int64_t at = row < a.count ? row : a.count - 1;
        if ((depDict >> d) & 1) {
          at = a.deps[d].indices[at];
        }
        dv[u][d] = ((dep32 >> d) & 1) ? static_cast<uint64_t>(static_cast<const uint32_t*>(a.deps[d].values)[at])
                                      : static_cast<const uint64_t*>(a.deps[d].values)[at];
      }
    }
#pragma unroll
    for (int u = 0; u < kAppendRows; ++u) {
      const int64_t row = base + u * 256LL;
      if (row >= a.count) {
        continue;
      }
      const int64_t out = a.base + row;
      if (a.keyNullOut) {
        a.keyNullOut[out] = 0;
      }
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        a.keyOut[k][out] = static_cast<uint64_t>(kv[u][k]);
        mn[k] = kv[u][k] < mn[k] ? kv[u][k] : mn[k];
        mx[k] = kv[u][k] > mx[k] ? kv[u][k] : mx[k];
      }
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        a.depValid[d][out] = 1;
        if ((dep32 >> d) & 1) {
          reinterpret_cast<uint32_t*>(a.depOut[d])[out] = static_cast<uint32_t>(dv[u][d]);
        } else {
          reinterpret_cast<uint64_t*>(a.depOut[d])[out] = dv[u][d];
        }
      }
    }
  }
  // one pair of atomics per key and WORKGROUP
  __shared__ int64_t waveLo[4][NK];
  __shared__ int64_t waveHi[4][NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    int64_t lo = mn[k], hi = mx[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int64_t olo = static_cast<int64_t>(shfl64(static_cast<uint64_t>(lo), lane() ^ off));
      const int64_t ohi = static_cast<int64_t>(shfl64(static_cast<uint64_t>(hi), lane() ^ off));
      lo = olo < lo ? olo : lo;
      hi = ohi > hi ? ohi : hi;
    }
    if (lane() == 0) {
      waveLo[threadIdx.x >> 6][k] = lo;
      waveHi[threadIdx.x >> 6][k] = hi;
    }
  }
  blockSync();
  if (threadIdx.x < NK) {
    const int k = threadIdx.x;
    int64_t lo = waveLo[0][k], hi = waveHi[0][k];
    for (int w = 1; w < 4; ++w) {
      lo = waveLo[w][k] < lo ? waveLo[w][k] : lo;
      hi = waveHi[w][k] > hi ? waveHi[w][k] : hi;
    }
    if (lo <= hi) {
      atomicMin(reinterpret_cast<long long*>(&a.counters->keyMin[k]), static_cast<long long>(lo));
      atomicMax(reinterpret_cast<long long*>(&a.counters->keyMax[k]), static_cast<long long>(hi));
    }
  }
}

using AppendFlatLauncher = void (*)(const AppendArgs&, uint32_t, uint32_t, uint32_t, uint32_t, int);
template <int NK, int ND>
void launchAppendFlat(const AppendArgs& a, uint32_t key32, uint32_t dep32, uint32_t keyDict, uint32_t depDict, int grid) {
  VX_LAUNCH("k_build_append", (k_build_append_flat<NK, ND>), grid, 256, 0, a, key32, dep32, keyDict, depDict);
}
constexpr int kAppendFlatKeys = 2;
constexpr int kAppendFlatDeps = 4;
const AppendFlatLauncher kAppendFlat[kAppendFlatKeys][kAppendFlatDeps + 1] = {
    {&launchAppendFlat<1, 0>, &launchAppendFlat<1, 1>, &launchAppendFlat<1, 2>, &launchAppendFlat<1, 3>,
     &launchAppendFlat<1, 4>},
    {&launchAppendFlat<2, 0>, &launchAppendFlat<2, 1>, &launchAppendFlat<2, 2>, &launchAppendFlat<2, 3>,
     &launchAppendFlat<2, 4>},
};

// true: the batch went through k_build_append_flat.
bool appendFlat(const AppendArgs& a) {
  if (a.rows != nullptr || a.keyValidWords != nullptr || a.nullAsValue || a.numKeys < 1 ||
      a.numKeys > kAppendFlatKeys || a.numDeps > kAppendFlatDeps) {
    return false;
  }
  uint32_t key32 = 0, dep32 = 0, keyDict = 0, depDict = 0;
  for (int k = 0; k < a.numKeys; ++k) {
    const ColView& c = a.keys[k];
    if ((c.enc != VX355_FLAT && c.enc != VX355_DICTIONARY) || c.nulls != nullptr || a.keyWords[k] != 1 ||
        (c.kind != VX355_INTEGER && c.kind != VX355_BIGINT)) {
      return false;
    }
    key32 |= c.kind == VX355_INTEGER ? 1u << k : 0u;
    keyDict |= c.enc == VX355_DICTIONARY ? 1u << k : 0u;
  }
  for (int d = 0; d < a.numDeps; ++d) {
    const ColView& c = a.deps[d];
    if ((c.enc != VX355_FLAT && c.enc != VX355_DICTIONARY) || c.nulls != nullptr ||
        (a.depWidth[d] != 4 && a.depWidth[d] != 8)) {
      return false;
    }
    dep32 |= a.depWidth[d] == 4 ? 1u << d : 0u;
    depDict |= c.enc == VX355_DICTIONARY ? 1u << d : 0u;
  }
  const int grid = static_cast<int>(std::max<int64_t>(
      1, std::min<int64_t>(ceilDiv(a.count, 256LL * kAppendRows), static_cast<int64_t>(Runtime::get().numCUs) * 8)));
  kAppendFlat[a.numKeys - 1][a.numDeps](a, key32, dep32, keyDict, depDict, grid);
  return true;
}

// ---- build: table -------------------------------------------------------------------
struct Slot {
  uint64_t key;
  uint32_t head;
  uint32_t pad;
};

// Slot with up to two 8-byte dependents of its (only) build row next to the key: a probe that finds
// the key has the payload in the same 32 bytes instead of gathering it by build row later (the
// reference's RowContainer keeps keys and dependents of a row side by side for the same reason).
// Built on demand for tables without duplicate keys when a probe batch is large enough to pay
// for the pass (vx355_join_table::wide).
constexpr int kWideDeps = 2;
struct WideSlot {
  uint64_t key;
  uint32_t head;
  uint32_t pad;
  uint64_t dep[kWideDeps];
};

// A table's slot array has ONE of the two layouts: slotShift 4 = Slot, 5 = WideSlot. Builds large
// enough to leave every cache are inserted into wide slots directly (buildFinish): the row that
// claims a key stores its dependents in the same 32 bytes, and no pass over the whole table
// (k_widen_slots) is needed before the first probe. Every kernel addresses slots through slotAt.
__device__ inline Slot* slotAt(Slot* base, uint64_t pos, int32_t shift) {
  return reinterpret_cast<Slot*>(reinterpret_cast<char*>(base) + (pos << shift));
}
__device__ inline const Slot* slotAt(const Slot* base, uint64_t pos, int32_t shift) {
  return reinterpret_cast<const Slot*>(reinterpret_cast<const char*>(base) + (pos << shift));
}

// Where a key's walk over the slots begins and how it goes on. homeMask = (capacity - 1) with the
// low bits cleared that number the slots of one 64-byte sector (4 narrow / 2 wide slots): the walk
// starts on a sector boundary, so its first slots cost ONE request to the L2 (a lookup is priced in
// requests: ~90 G/s per chip however they are spread, profiles/r06_c5_counters.md). wrapMask: the walk
// wraps inside the group of (wrapMask + 1) slots it started in - the whole table for tables
// inserted slot by slot, 128 KB of slots for tables assembled group by group in LDS (k_lds_build).
__device__ inline uint64_t nextSlot(uint64_t pos, uint64_t wrapMask) { return (pos & ~wrapMask) | ((pos + 1) & wrapMask); }

__global__ __launch_bounds__(256) void k_widen_slots(const Slot* slots, WideSlot* wide, uint64_t capacity,
                                                     const uint64_t* dep0, const uint64_t* dep1) {
  for (uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; i < capacity;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const Slot s = slots[i];
    WideSlot w;
    w.key = s.key;
    w.head = s.head;
    w.pad = 0;
    const bool live = s.key != kEmptyKey;
    w.dep[0] = live && dep0 ? dep0[s.head] : 0;
    w.dep[1] = live && dep1 ? dep1[s.head] : 0;
    wide[i] = w;
  }
}

struct InsertArgs {
  const uint64_t* keyStore[kMaxKeys];
  int32_t keyWords[kMaxKeys];
  int32_t keyIsString[kMaxKeys];
  int32_t keyKind[kMaxKeys];
  KeyRange ranges[kMaxKeys];
  int32_t numKeys;
  int32_t mode;
  int64_t numRows;
  uint32_t* head;     // array mode (NOT initialised: 'present' says which entries are live)
  uint32_t* present;  // array mode: one bit per possible key
  Slot* slots;        // normalized-key mode (slotShift 4: Slot, 5: WideSlot)
  int32_t slotShift;
  int32_t pad0;
  uint64_t homeMask, wrapMask;         // see nextSlot
  const uint64_t* wideDep[kWideDeps];  // slotShift 5: the dependents the claiming row stores next to its key
  uint64_t* gslots;   // generic hash mode: {hash tag:32 | representative row + 1:32}
  uint64_t capacity;
  uint32_t* next;
  int32_t phase;      // array mode: 1 = claim keys, 2 = chain the duplicates
  // HashJoinNode::canDropDuplicates (core/PlanNode.h:3391-3398; HashBuild.cpp:517-548): a semi / anti
  // join without an extra filter only asks whether a key exists - a row whose key is already in
  // the table is not linked at all (no chain, no duplicate count: the table behaves like one with
  // unique keys and the probe never walks a chain)
  int32_t dropDups;
  BuildCounters* counters;
  const uint8_t* keyNull;  // rows kept for right / full joins only: not inserted
  int32_t nullAsValue;     // keyNull[row] is a mask of null keys instead, and such rows are inserted
  int32_t pad2;
};

constexpr uint32_t kPendingRow = 0xfffffffeu;

// int64 image VectorHasher normalises, from the stored key image.
__device__ inline int64_t storedKeyValue(const uint64_t* store, int32_t words, int32_t isString, int64_t row) {
  if (!isString) {
    return static_cast<int64_t>(store[row * words]);
  }
  StringView16 v;
  const uint64_t w0 = store[row * 2];
  v.size = static_cast<uint32_t>(w0);
  v.prefix = static_cast<uint32_t>(w0 >> 32);
  v.tail = store[row * 2 + 1];
  return stringAsNumber(v);
}

__device__ inline uint64_t buildKey(const InsertArgs& a, int64_t row) {
  uint64_t key = 0;
  for (int k = 0; k < a.numKeys; ++k) {
    if (a.nullAsValue && a.keyNull && ((a.keyNull[row] >> k) & 1)) {
      continue;  // value id 0 = null (VectorHasher.h:188)
    }
    const int64_t v = storedKeyValue(a.keyStore[k], a.keyWords[k], a.keyIsString[k], row);
    const uint64_t id = static_cast<uint64_t>(v) - static_cast<uint64_t>(a.ranges[k].min) + 1;
    key += a.ranges[k].multiplier * id;
  }
  return key;
}

__device__ inline bool storedKeysEqual(const InsertArgs& a, int64_t r1, int64_t r2) {
  if (a.nullAsValue && a.keyNull && a.keyNull[r1] != a.keyNull[r2]) {
    return false;  // null equals null only (the images of null keys are zero)
  }
  for (int k = 0; k < a.numKeys; ++k) {
    const int w = a.keyWords[k];
    if (a.keyIsString[k]) {
      if (!stringImagesEqual(a.keyStore[k][r1 * 2], a.keyStore[k][r1 * 2 + 1], a.keyStore[k][r2 * 2],
                             a.keyStore[k][r2 * 2 + 1])) {
        return false;
      }
      continue;
    }
    if (a.keyStore[k][r1 * w] != a.keyStore[k][r2 * w]) {
      return false;
    }
    if (w == 2 && a.keyStore[k][r1 * 2 + 1] != a.keyStore[k][r2 * 2 + 1]) {
      return false;
    }
  }
  return true;
}

// VectorHasher hash of a build row's keys from their stored images (generic hash mode).
__device__ inline uint64_t storedRowHash(const InsertArgs& a, int64_t row) {
  uint64_t hash = 0;
  for (int k = 0; k < a.numKeys; ++k) {
    const int w = a.keyWords[k];
    uint64_t hv = hashFromImage(a.keyKind[k], a.keyStore[k][row * w], w == 2 ? a.keyStore[k][row * 2 + 1] : 0);
    if (a.nullAsValue && a.keyNull && ((a.keyNull[row] >> k) & 1)) {
      hv = kNullHash;
    }
    hash = k == 0 ? hv : hashMix(hash, hv);
  }
  return hash;
}

// Sets the presence bit of 'key' for every active lane of the wave and tells each
// lane whether it was the first to claim its key. HBM atomics retire at ~20 G/s
// chip-wide, and build sides usually arrive in key order (a scan of the dimension
// table): consecutive lanes whose keys share a 32-bit word of the bitmap are
// combined (segmented OR scan over the run) and issue ONE atomicOr per run.
// Called by all 64 lanes.
// Split in two so that a lane can have several claims in flight: the atomic's return value is
// only consumed by resolveClaim. (Measured on the Q3 build: no gain - that insert is bound by
// the 14.6 M scattered 4-byte head stores, one 32-byte sector each, which the fabric retires at
// the ~30 G/s of tools/atomic_bench.hip's plain read-modify-write line.)
struct PresenceClaim {
  uint32_t prev;  // the tail lane's atomicOr result
  uint32_t incl;  // OR of the bits of this lane's run up to and including this lane
  uint32_t bit;
  int myTail;
  bool head;
  bool active;
};

__device__ inline PresenceClaim issueClaim(uint32_t* present, bool active, uint64_t key) {
  PresenceClaim c;
  const int ln = lane();
  const uint64_t word = active ? (key >> 5) : ~static_cast<uint64_t>(ln);  // inactive lanes: runs of their own
  c.bit = active ? 1u << (key & 31) : 0u;
  c.active = active;
  const uint64_t prevLaneWord = shfl64(word, ln > 0 ? ln - 1 : 0);
  c.head = ln == 0 || prevLaneWord != word;
  const uint64_t heads = ballot(c.head);
  const int run = static_cast<int>(popc64(heads & ((2ULL << ln) - 1)));  // 1-based run id, ascending
  uint32_t incl = c.bit;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = __shfl_up(incl, off, kWave);
    const int r = __shfl_up(run, off, kWave);
    if (ln >= off && r == run) {
      incl |= v;
    }
  }
  c.incl = incl;
  const bool tail = ln == 63 || ((heads >> (ln + 1)) & 1);
  const uint64_t tails = ballot(tail);
  c.myTail = ln + __ffsll(static_cast<long long>(tails >> ln)) - 1;
  c.prev = 0;
  if (tail && active) {
    c.prev = atomicOr(present + word, incl);
  }
  return c;
}

__device__ inline bool resolveClaim(const PresenceClaim& c) {
  const uint32_t prev = __shfl(c.prev, c.myTail, kWave);
  const uint32_t before = __shfl_up(c.incl, 1, kWave);
  const uint32_t earlier = c.head ? 0u : before;
  return c.active && !((prev | earlier) & c.bit);
}

__device__ inline bool claimPresence(uint32_t* present, bool active, uint64_t key) {
  return resolveClaim(issueClaim(present, active, key));
}

// Array mode never initialises the (possibly multi-GB) head array: phase 1
// claims each key with one atomicOr on a presence bitmap that is 32x smaller
// than the head array (and is what the probe consults first), the winner
// stores its row as chain head; phase 2, launched only when duplicates exist,
// pushes the other rows in front (HashTable.cpp:1394-1409 arrayPushRow).
__global__ __launch_bounds__(256) void k_join_insert(InsertArgs a) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  uint32_t dups = 0, distinct = 0;
  if (a.mode == JMODE_ARRAY && a.phase == 1) {
    // whole waves iterate together (the claims are wave-wide operations); four rows per lane in
    // flight, so four atomics of a wave overlap their round trips
    constexpr int kInFlight = 4;
    const int64_t rounds = (a.numRows + stride - 1) / stride;
    int64_t row0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    for (int64_t r = 0; r < rounds; r += kInFlight, row0 += stride * kInFlight) {
      PresenceClaim claim[kInFlight];
      uint64_t keys[kInFlight];
      bool nullKey[kInFlight];
#pragma unroll
      for (int u = 0; u < kInFlight; ++u) {
        const int64_t row = row0 + u * stride;
        const bool inRange = r + u < rounds && row < a.numRows;
        nullKey[u] = inRange && !a.nullAsValue && a.keyNull && a.keyNull[row];
        const bool active = inRange && !nullKey[u];
        keys[u] = active ? buildKey(a, row) : 0;
        claim[u] = issueClaim(a.present, active, keys[u]);
      }
#pragma unroll
      for (int u = 0; u < kInFlight; ++u) {
        const int64_t row = row0 + u * stride;
        const bool first = resolveClaim(claim[u]);
        if (nullKey[u]) {
          a.next[row] = kNoRow32;
        } else if (claim[u].active) {
          if (first) {
            a.head[keys[u]] = static_cast<uint32_t>(row);
            a.next[row] = kNoRow32;
            ++distinct;
          } else if (a.dropDups) {
            a.next[row] = kNoRow32;
          } else {
            a.next[row] = kPendingRow;
            ++dups;
          }
        }
      }
    }
    // one add per wave instead of one per lane
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      dups += __shfl_xor(dups, off, kWave);
      distinct += __shfl_xor(distinct, off, kWave);
    }
    if (lane() == 0 && dups) {
      atomicAdd(&a.counters->duplicates, dups);
    }
    if (lane() == 0 && distinct) {
      atomicAdd(&a.counters->numDistinct, distinct);
    }
    return;
  }
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; row < a.numRows;
       row += stride) {
    if (!a.nullAsValue && a.keyNull && a.keyNull[row]) {
      if (a.phase == 1) {
        a.next[row] = kNoRow32;
      }
      continue;
    }
    if (a.mode == JMODE_ARRAY) {
      if (a.next[row] == kPendingRow) {
        const uint64_t key = buildKey(a, row);
        a.next[row] = atomicExch(a.head + key, static_cast<uint32_t>(row));
      }
      continue;
    }
    if (a.mode == JMODE_HASH) {
      // Generic keys (the reference's kHash): slot = {hash tag, representative
      // row}. The key images were stored by an earlier launch, so a claimed slot
      // is immediately comparable; equal keys are pushed behind the
      // representative (pushNext, HashTable.cpp:1412-1418); next[] was
      // pre-filled with "no row".
      const uint64_t hash = storedRowHash(a, row);
      const uint64_t tag = hash >> 32;
      const uint64_t gmask = a.capacity - 1;
      uint64_t pos = slotOfHash(hash, gmask);
      bool placed = false;
      for (uint64_t probes = 0; probes <= gmask && !placed; ++probes) {
        unsigned long long w = __hip_atomic_load(a.gslots + pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == 0) {
          const unsigned long long mine = (tag << 32) | (static_cast<uint64_t>(row) + 1);
          const unsigned long long old =
              atomicCAS(reinterpret_cast<unsigned long long*>(a.gslots + pos), 0ULL, mine);
          if (old == 0) {
            ++distinct;
            placed = true;
            break;
          }
          w = old;
        }
        if ((w >> 32) == tag) {
          const int64_t rep = static_cast<int64_t>(static_cast<uint32_t>(w)) - 1;
          if (storedKeysEqual(a, row, rep)) {
            if (!a.dropDups) {
              a.next[row] = atomicExch(a.next + rep, static_cast<uint32_t>(row));
              ++dups;
            }
            placed = true;
            break;
          }
        }
        pos = (pos + 1) & gmask;
      }
      if (!placed) {
        a.counters->tableFull = 1;
      }
      continue;
    }
    // Normalized-key mode. Random read-modify-writes retire at ~20 G/s on this chip whatever their
    // locality (profiles/r02_atomic_bench.txt), so an insert is priced in atomics: phase 1 is ONE
    // compare-and-swap per row on the slot's key - the row that claims the key fills the rest of
    // the slot with plain stores (nobody reads a head in this launch), a row that finds its key
    // already there waits for phase 2, launched only when such rows exist, which pushes it in
    // front of the chain (pushNext, HashTable.cpp:1412-1418) with one exchange on the head.
    const uint64_t key = buildKey(a, row);
    const uint64_t mask = a.capacity - 1;
    uint64_t pos = twangMix64(key) & a.homeMask;
    if (a.phase == 2) {
      if (a.next[row] != kPendingRow) {
        continue;
      }
      for (uint64_t probes = 0; probes <= mask; ++probes) {
        Slot* s = slotAt(a.slots, pos, a.slotShift);
        if (s->key == key) {
          a.next[row] = atomicExch(&s->head, static_cast<uint32_t>(row));
          break;
        }
        pos = nextSlot(pos, a.wrapMask);
      }
      continue;
    }
    bool placed = false;
    for (uint64_t probes = 0; probes <= mask; ++probes) {
      Slot* s = slotAt(a.slots, pos, a.slotShift);
      const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&s->key), kEmptyKey, key);
      if (old == kEmptyKey) {
        s->head = static_cast<uint32_t>(row);
        if (a.slotShift == 5) {
          // the claiming row's dependents ride in the slot (read only when the table turns out
          // to hold no duplicate keys)
          WideSlot* w = reinterpret_cast<WideSlot*>(s);
          w->dep[0] = a.wideDep[0] ? a.wideDep[0][row] : 0;
          w->dep[1] = a.wideDep[1] ? a.wideDep[1][row] : 0;
        }
        a.next[row] = kNoRow32;
        ++distinct;
        placed = true;
        break;
      }
      if (old == key) {
        if (a.dropDups) {
          a.next[row] = kNoRow32;  // the claiming row stays the key's only row
        } else {
          a.next[row] = kPendingRow;
          ++dups;
        }
        placed = true;
        break;
      }
      pos = nextSlot(pos, a.wrapMask);
    }
    if (!placed) {
      a.counters->tableFull = 1;
    }
  }
  // the loops above leave the wave converged here: one add per wave (see phase 1)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    dups += __shfl_xor(dups, off, kWave);
    distinct += __shfl_xor(distinct, off, kWave);
  }
  if (lane() == 0 && dups) {
    atomicAdd(&a.counters->duplicates, dups);
  }
  if (lane() == 0 && distinct) {
    atomicAdd(&a.counters->numDistinct, distinct);
  }
}

__global__ __launch_bounds__(256) void k_fill_u32(uint32_t* p, uint64_t n, uint32_t v) {
  const uint64_t step = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += step) {
    p[i] = v;
  }
}

// Counting joins: remaining[head of the row's chain] += 1 for every inserted build row
// (the per-key count HashBuild keeps while it deduplicates, HashBuild.cpp:534-548).
__global__ __launch_bounds__(256) void k_count_init(InsertArgs a, uint32_t* remaining) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; row < a.numRows;
       row += stride) {
    uint32_t head = kNoRow32;
    if (a.mode == JMODE_HASH) {
      // the chain hangs off the slot's representative row, which is what a probe's hits[] names
      const uint64_t hash = storedRowHash(a, row);
      const uint64_t tag = hash >> 32;
      const uint64_t mask = a.capacity - 1;
      uint64_t pos = slotOfHash(hash, mask);
      for (uint64_t probes = 0; probes <= mask; ++probes) {
        const uint64_t w = a.gslots[pos];
        if (w == 0) {
          break;
        }
        if ((w >> 32) == tag) {
          const int64_t candidate = static_cast<int64_t>(static_cast<uint32_t>(w)) - 1;
          if (storedKeysEqual(a, row, candidate)) {
            head = static_cast<uint32_t>(candidate);
            break;
          }
        }
        pos = (pos + 1) & mask;
      }
    } else if (a.mode == JMODE_ARRAY) {
      head = a.head[buildKey(a, row)];
    } else {
      const uint64_t key = buildKey(a, row);
      const uint64_t mask = a.capacity - 1;
      uint64_t pos = twangMix64(key) & a.homeMask;
      for (uint64_t probes = 0; probes <= mask; ++probes) {
        const Slot sl = *slotAt(a.slots, pos, a.slotShift);
        if (sl.key == key) {
          head = sl.head;
          break;
        }
        if (sl.key == kEmptyKey) {
          break;
        }
        pos = nextSlot(pos, a.wrapMask);
      }
    }
    if (head != kNoRow32) {
      atomicAdd(remaining + head, 1u);
    }
  }
}

// n 16-byte units; wide slots (two units each) get {empty key, no row} in the even unit, zeros in the odd one.
__global__ __launch_bounds__(256) void k_fill_slots(Slot* p, uint64_t n, int32_t wide) {
  const uint64_t step = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += step) {
    Slot s;
    const bool keyUnit = !wide || (i & 1) == 0;
    s.key = keyUnit ? kEmptyKey : 0;
    s.head = keyUnit ? kNoRow32 : 0;
    s.pad = 0;
    p[i] = s;
  }
}

// ---- probe -------------------------------------------------------------------------
constexpr int kTileRows = 8192;  // probe rows per workgroup tile (256 lanes x 32)
constexpr int kProbeUnroll = 4;      // probes per lane in flight
#ifndef VX355_PROBE_UNROLL_FAST
#define VX355_PROBE_UNROLL_FAST 8
#endif
constexpr int kProbeUnrollFast = VX355_PROBE_UNROLL_FAST;  // ... of the flat BIGINT key paths

// FilterProject -> HashProbe fusion (vx355_join_probe_set_input_filter): the conjunction the
// FilterProject in front of the probe would have applied, evaluated on the probe's own input rows.
// A row that fails is a miss (the fusion is only offered for join kinds whose misses emit nothing).
struct RowFilter {
  TermArg terms[kMaxTerms];
  int32_t numTerms;
  // The common shape - ONE term over a flat INTEGER / DATE / BIGINT column without nulls against an
  // integer constant (TPC-H Q3: l_shipdate > d, o_orderdate < d) - as a range test the flat-key
  // probe paths evaluate on values they prefetch next to the keys: pass = (lo <= v <= hi) != invert.
  int32_t fast;          // 0 = use terms[]; 4 / 8 = byte width of the fast column
  const void* fastValues;
  int64_t lo, hi;
  int32_t invert;
  int32_t pad;
};
// The raw word of the fast filter's column (W = 4 or 8 bytes). NOT widened here: a conversion right
// behind the load needs the loaded value and would make every load wait for itself (s_waitcnt vmcnt(0)
// behind each global_load: the eight rows of an iteration then load one after the other).
template <int W>
struct FastWord {
  using type = int64_t;
};
template <>
struct FastWord<4> {
  using type = int32_t;
};
template <int W>
__device__ inline typename FastWord<W>::type fastFilterLoad(const RowFilter& f, int64_t row) {
  return __builtin_nontemporal_load(static_cast<const typename FastWord<W>::type*>(f.fastValues) + row);
}
__device__ inline bool fastFilterPass(const RowFilter& f, int64_t v) {
  return ((v >= f.lo) & (v <= f.hi)) != (f.invert != 0);
}
__device__ inline bool rowPasses(const RowFilter& f, int64_t row) {
  return evalFilter(f.terms, f.numTerms, row);
}
// (32-bit column: lo / hi were clamped to the int32 range by the host, an empty range is lo > hi)
__device__ inline bool fastFilterPass(const RowFilter& f, int32_t v) {
  return ((v >= static_cast<int32_t>(f.lo)) & (v <= static_cast<int32_t>(f.hi))) != (f.invert != 0);
}

struct ProbeArgs {
  ColView keys[kMaxKeys];
  KeyRange ranges[kMaxKeys];
  int32_t numKeys;
  int32_t mode;
  int32_t hasDuplicates;
  int32_t joinType;
  int64_t numRows;
  int64_t numTiles;
  const uint32_t* head;
  const uint32_t* present;
  const Slot* slots;
  int32_t slotShift;                   // 4: Slot, 5: WideSlot (slots == wide)
  int32_t pad4;
  uint64_t homeMask, wrapMask;         // see nextSlot
  const uint64_t* gslots;              // generic hash mode
  const uint64_t* keyStore[kMaxKeys];  // generic hash mode: build key images
  int32_t keyWords[kMaxKeys];
  uint64_t capacity;
  const uint32_t* next;
  uint32_t* hits;       // first matching build row or kNoRow32
  uint32_t* counts;     // output rows per probe row; only kept for duplicate tables
  uint64_t* tileSums;   // output rows per tile
  uint32_t* unitSums;   // grouped probe with units of a quarter tile: output rows per unit (k_emit skips its count pass)
  int32_t fastKey;      // single non-null BIGINT key (FK of TPC-H joins): 1 flat, 2 dictionary wrapped
  int32_t nullAware;    // null-aware anti join on a non-empty build side: null probe keys produce nothing
  uint8_t* probed;      // right / full / right semi: build rows some probe row matched
  // Sparse listing (low hit rates): the probe pass itself lists the hits of every
  // tile, in probe-row order, into staged[tile * kSparseCap ...]; hits[] is only
  // written for the tiles that overflow the staging segment (tileDense).
  int64_t tileBegin;    // tiles [tileBegin, numTiles) of this launch
  uint2* staged;        // {probe row, build row}
  uint8_t* tileDense;
  int32_t countHits;    // add every tile's hit count to sparseStats[1] (the sample launch)
  int32_t pad3;
  uint64_t* sparseStats;  // [0] overflowed tiles, [1] hits listed (HBM: atomics on the pinned mailbox cross PCIe)
  int32_t nullAsValue;            // a null probe key is a value: id 0 / kNullHash
  int32_t window;                 // listing probe, array mode, flat BIGINT key: presence bits through a per-wave window
  const uint8_t* keyNullStore;    // generic hash mode + nullAsValue: null-key mask per build row
  uint64_t presentWords;          // u32 words of the presence bitmap
  const WideSlot* wide;           // WIDE probes: slots with inline dependents ...
  uint64_t* hitVals[kWideDeps];   // ... whose values are staged per probe row next to hits[]
  RowFilter rf;                   // fused input filter (numTerms == 0: none)
};

constexpr int kSparseCap = 1024;  // staged pairs per tile of 8192 probe rows (12.5 % hit rate)

constexpr uint32_t kNullKey32 = 0xfffffffeu;  // hits[]: the probe key holds a null (null-aware anti only)

__device__ inline bool isHit(uint32_t hit) { return hit < kNullKey32; }

// Output rows of one probe row with 'matches' matching build rows.
__host__ __device__ inline uint32_t outputCount(int32_t joinType, uint32_t matches, bool nullKey = false) {
  switch (joinType) {
    case VX355_JOIN_INNER:
    case VX355_JOIN_RIGHT:
      return matches;
    case VX355_JOIN_LEFT:
    case VX355_JOIN_FULL:
      return matches ? matches : 1;
    case VX355_JOIN_LEFT_SEMI_FILTER:
      return matches ? 1 : 0;
    case VX355_JOIN_LEFT_SEMI_PROJECT:
      return 1;
    case VX355_JOIN_RIGHT_SEMI_FILTER:
    case VX355_JOIN_RIGHT_SEMI_PROJECT:
    case VX355_JOIN_RIGHT_ANTI:
      return 0;  // build-side output only (processRightSemiNoFilter, HashProbe.cpp:1422-1429)
    case VX355_JOIN_COUNTING_LEFT_SEMI_FILTER:
      return matches ? 1 : 0;  // after k_count_consume: a hit = the row consumed an occurrence
    default:  // ANTI: rows without a match; null keys included unless null aware
      return (matches || nullKey) ? 0 : 1;
  }
}

__host__ __device__ inline bool listsMatches(int32_t joinType) {
  return joinType == VX355_JOIN_INNER || joinType == VX355_JOIN_LEFT || joinType == VX355_JOIN_RIGHT ||
      joinType == VX355_JOIN_FULL;
}

// Normalized key of a probe row with lookupValueIds semantics: false = proven
// miss (null key, or a value outside the build side's range).
__device__ inline bool probeKey(const ProbeArgs& a, int64_t row, uint64_t* keyOut) {
  uint64_t key = 0;
  for (int k = 0; k < a.numKeys; ++k) {
    const ColView& c = a.keys[k];
    if (colIsNull(c, row)) {
      if (a.nullAsValue) {
        continue;  // value id 0
      }
      return false;  // a null key never matches (HashProbe.cpp:778)
    }
    int64_t v;
    bool mappable;
    const uint64_t id = valueIdAt(c, colIndex(c, row), a.ranges[k], &v, &mappable);
    if (id == 0) {
      return false;
    }
    key += a.ranges[k].multiplier * id;
  }
  *keyOut = key;
  return true;
}

// Generic mode probe: VectorHasher hash of the probe row, tag filter, key images.
__device__ inline uint32_t lookupGeneric(const ProbeArgs& a, int64_t row) {
  uint64_t w0[kMaxKeys], w1[kMaxKeys];
  uint64_t hash = 0;
  uint32_t nullMask = 0;
#pragma unroll
  for (int k = 0; k < kMaxKeys; ++k) {
    w0[k] = 0;
    w1[k] = 0;
    if (k < a.numKeys) {
      const ColView& c = a.keys[k];
      if (colIsNull(c, row)) {
        if (!a.nullAsValue) {
          return kNoRow32;
        }
        nullMask |= 1u << k;
        hash = k == 0 ? kNullHash : hashMix(hash, kNullHash);
        continue;
      }
      const int64_t i = colIndex(c, row);
      bool inlineOk = true;
      keyImage(c, i, &w0[k], &w1[k], &inlineOk);  // non-inline strings: {size | prefix, pointer}
      const uint64_t hv = hashValueAt(c, i);
      hash = k == 0 ? hv : hashMix(hash, hv);
    }
  }
  const uint64_t tag = hash >> 32;
  const uint64_t mask = a.capacity - 1;
  uint64_t pos = slotOfHash(hash, mask);
  for (uint64_t probes = 0; probes <= mask; ++probes) {
    const uint64_t w = a.gslots[pos];
    if (w == 0) {
      return kNoRow32;
    }
    if ((w >> 32) == tag) {
      const uint64_t rep = static_cast<uint64_t>(static_cast<uint32_t>(w)) - 1;
      bool equal = !a.nullAsValue || (a.keyNullStore ? a.keyNullStore[rep] : 0) == nullMask;
#pragma unroll
      for (int k = 0; k < kMaxKeys; ++k) {
        if (equal && k < a.numKeys) {
          if (a.keys[k].kind == VX355_VARCHAR || a.keys[k].kind == VX355_VARBINARY) {
            equal = stringImagesEqual(w0[k], w1[k], a.keyStore[k][rep * 2], a.keyStore[k][rep * 2 + 1]);
          } else {
            equal = a.keyStore[k][rep * a.keyWords[k]] == w0[k];
            if (equal && a.keyWords[k] == 2) {
              equal = a.keyStore[k][rep * 2 + 1] == w1[k];
            }
          }
        }
      }
      if (equal) {
        return static_cast<uint32_t>(rep);
      }
    }
    pos = (pos + 1) & mask;
  }
  return kNoRow32;
}

__device__ inline uint32_t lookupSlots(const ProbeArgs& a, uint64_t key) {
  const uint64_t mask = a.capacity - 1;
  uint64_t pos = twangMix64(key) & a.homeMask;
  for (uint64_t probes = 0; probes <= mask; ++probes) {
    const uint4 raw = *reinterpret_cast<const uint4*>(slotAt(a.slots, pos, a.slotShift));
    const uint64_t k = (static_cast<uint64_t>(raw.y) << 32) | raw.x;
    if (k == key) {
      return raw.z;
    }
    if (k == kEmptyKey) {
      return kNoRow32;
    }
    pos = nextSlot(pos, a.wrapMask);
  }
  return kNoRow32;
}

template <int WIDE>
__device__ inline uint32_t lookupWide(const ProbeArgs& a, uint64_t key, uint64_t (&vals)[kWideDeps]) {
  const uint64_t mask = a.capacity - 1;
  uint64_t pos = twangMix64(key) & a.homeMask;
  for (uint64_t probes = 0; probes <= mask; ++probes) {
    const uint4* at = reinterpret_cast<const uint4*>(a.wide + pos);
    const uint4 raw = at[0];
    const uint4 dep = at[1];   // same 32-byte sector as the key
    const uint64_t k = (static_cast<uint64_t>(raw.y) << 32) | raw.x;
    if (k == key) {
      vals[0] = (static_cast<uint64_t>(dep.y) << 32) | dep.x;
      if (WIDE > 1) {
        vals[1] = (static_cast<uint64_t>(dep.w) << 32) | dep.z;
      }
      return raw.z;
    }
    if (k == kEmptyKey) {
      return kNoRow32;
    }
    pos = nextSlot(pos, a.wrapMask);
  }
  return kNoRow32;
}

// The KU lookups of one lane, first sectors first: the 64-byte sector every key's walk begins in
// (4 narrow / 2 wide slots, see nextSlot) is loaded for four of the lane's keys at a time before any
// is looked at - 16 independent 16-byte loads in flight, and since the four loads of one sector
// merge into one request, one L2 request per key unless its walk leaves the sector (a walk per
// key, slot by slot, keeps ONE load in flight and pays a request per slot). joinNormalizedKeyProbe
// prefetches the 64 buckets of its window the same way before it compares
// (exec/HashTable.cpp:697-725).
template <int WIDE, int KU>
__device__ inline void lookupFirstSlots(const ProbeArgs& a, const uint64_t (&key)[KU], const bool (&candidate)[KU],
                                        uint32_t (&hit)[KU], uint64_t (*vals)[kWideDeps]) {
  const uint64_t mask = a.capacity - 1;
  const char* base = WIDE > 0 ? reinterpret_cast<const char*>(a.wide) : reinterpret_cast<const char*>(a.slots);
  const int shift = WIDE > 0 ? 5 : a.slotShift;
  constexpr int H = KU > 4 ? 4 : KU;
  static_assert(KU % H == 0, "rounds of four keys");
#pragma unroll
  for (int h0 = 0; h0 < KU; h0 += H) {
    uint64_t home[H];
    uint4 q[H][4];
#pragma unroll
    for (int u = 0; u < H; ++u) {
      home[u] = twangMix64(key[h0 + u]) & a.homeMask;
    }
#pragma unroll
    for (int u = 0; u < H; ++u) {
      // (rows that are no candidates read sector 0: unconditional loads stay back to back)
      const uint4* at = reinterpret_cast<const uint4*>(base + ((candidate[h0 + u] ? home[u] : 0) << shift));
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        q[u][w] = at[w];
      }
    }
#pragma unroll
    for (int u = 0; u < H; ++u) {
      const int g = h0 + u;
      hit[g] = kNoRow32;
      if (!candidate[g]) {
        continue;
      }
      bool done = false;
      const int inSector = 64 >> shift;
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        if (done || sl >= inSector) {
          continue;
        }
        const uint4 kq = shift == 5 ? q[u][(2 * sl) & 3] : q[u][sl];
        const uint64_t k = (static_cast<uint64_t>(kq.y) << 32) | kq.x;
        if (k == key[g]) {
          hit[g] = kq.z;
          if constexpr (WIDE > 0) {
            const uint4 d = q[u][(2 * sl + 1) & 3];
            vals[g][0] = (static_cast<uint64_t>(d.y) << 32) | d.x;
            if (WIDE > 1) {
              vals[g][1] = (static_cast<uint64_t>(d.w) << 32) | d.z;
            }
          }
          done = true;
        } else if (k == kEmptyKey) {
          done = true;
        }
      }
      if (!done) {
        // the sector is full of other keys: walk on from its last slot
        uint64_t at = nextSlot(home[u] + static_cast<uint64_t>(inSector) - 1, a.wrapMask);
        for (uint64_t probes = 1; probes <= mask; ++probes) {
          const uint4* s = reinterpret_cast<const uint4*>(base + (at << shift));
          const uint4 r = s[0];
          const uint64_t k2 = (static_cast<uint64_t>(r.y) << 32) | r.x;
          if (k2 == key[g]) {
            hit[g] = r.z;
            if constexpr (WIDE > 0) {
              const uint4 d = s[1];
              vals[g][0] = (static_cast<uint64_t>(d.y) << 32) | d.x;
              if (WIDE > 1) {
                vals[g][1] = (static_cast<uint64_t>(d.w) << 32) | d.z;
              }
            }
            break;
          }
          if (k2 == kEmptyKey) {
            break;
          }
          at = nextSlot(at, a.wrapMask);
        }
      }
    }
  }
}

// HashTable::joinProbe: hits[row] = first build row with an equal key. One
// workgroup per tile of 8192 consecutive probe rows; each lane keeps
// kProbeUnroll independent probes in flight: all key loads, then all presence
// bits (a bitmap 32x smaller than the head array, Infinity-Cache resident for
// TPC-H sized builds), then the head words of the survivors only. The tile's
// output-row count falls out of the same pass (listJoinResults needs it).
// MODE / FAST >= 0 fix a.mode / a.fastKey at compile time (the array-mode fast
// paths then need half the registers of the all-purpose instantiation <-1, -1>).
//
// SPARSE (listJoinResultsFastPath for joins that emit matches only and whose
// chains have one row, HashTable.cpp:2293-2350): instead of writing hits[] for
// every probe row and scanning it again at output time, every wave owns 2048
// consecutive rows of the tile and appends its hits {row, build row} to its own
// LDS list in row order (ballot + lane prefix, no atomics); after one barrier
// the four lists are written out back to back, i.e. in probe-row order. A tile
// in which some wave finds more than kSparseWaveCap hits falls back to the
// dense form.
constexpr int kSparseWaveCap = kSparseCap / 4;  // staged pairs per wave (2048 probe rows)

struct SparseLds {
  unsigned long long list[4][kSparseWaveCap];  // per wave, in probe-row order
  uint32_t waveCount[2][4];                    // double buffered by tile parity: one barrier per tile
};

// Key loads of the flat BIGINT probe paths: every key is read once, so it streams past the caches
// (nontemporal) and leaves the L2 to the presence bitmap / the slots - k_join_probe_list 0.67 -> 0.64 ms
// per 323 M probes of TPC-H Q3 on the same box. -DVX355_PROBE_CACHED_KEYS restores plain loads.
#ifdef VX355_PROBE_CACHED_KEYS
#define VX355_KEY_LOAD(p) (*(p))
#else
#define VX355_KEY_LOAD(p) __builtin_nontemporal_load(p)
#endif

// RF: 0 = no input filter at all (the instantiations every plain probe runs: not an instruction of the
// fusion in them); 4 / 8 = byte width of the fast filter's column (RowFilter::fast); -1 = a.rf.terms through
// evalFilter (any other filter shape: only the all-purpose instantiation <-1, -1> is built with it)
template <int MODE, int FAST, bool SPARSE, int WIDE = 0, int RF = 0, int THREADS = 256, int ROWS = kTileRows>
__device__ inline uint64_t probeTileBody(const ProbeArgs& a, int64_t tile, SparseLds* lds) {
  const int mode = MODE >= 0 ? MODE : a.mode;
  const int fastKey = FAST >= 0 ? FAST : a.fastKey;
  // probes per lane in flight: the flat BIGINT key paths stream the key column and are bound by
  // the bytes they keep in flight (8 per lane: 0.63 ms per 323 M probes, 4: 0.76 ms); the other
  // paths carry more state per probe
  constexpr int kU = FAST == 1 ? kProbeUnrollFast : kProbeUnroll;
  const int64_t tileBase = tile * ROWS;
  uint64_t mine = 0;
  static_assert(!SPARSE || (THREADS == 256 && ROWS == kTileRows), "the listing form is written for four waves and whole tiles");
  constexpr int kIters = ROWS / (THREADS * kU);
  static_assert(kIters >= 1, "a tile holds at least one round of the workgroup");
  auto rowOf = [&](int it, int u) -> int64_t {
    return SPARSE ? tileBase + (threadIdx.x >> 6) * (kTileRows / 4) + (it * kU + u) * 64 + lane()
                  : tileBase + (it * kU + u) * THREADS + threadIdx.x;
  };
  // Flat BIGINT key (FAST == 1): the key loads of iteration it + 1 are issued behind the
  // bitmap gathers of iteration it, so the HBM latency of the keys overlaps the cache
  // latency of the dependent gathers instead of adding to it.
  int64_t vnext[kU];
  typename FastWord<RF>::type fnext[RF > 0 ? kU : 1];  // RF > 0: the fused filter's column, prefetched like the keys
  if (FAST == 1) {
    const int64_t* kp = static_cast<const int64_t*>(a.keys[0].values);
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t r = rowOf(0, u);
      vnext[u] = VX355_KEY_LOAD(kp + (r < a.numRows ? r : a.numRows - 1));
      if constexpr (RF > 0) {
        fnext[u] = fastFilterLoad<RF>(a.rf, r < a.numRows ? r : a.numRows - 1);
      }
    }
  }
  for (int it = 0; it < kIters; ++it) {
    int64_t rows[kU];
    uint64_t key[kU];
    bool candidate[kU];
    uint32_t hit[kU];
    uint64_t vals[WIDE > 0 && !SPARSE ? kU : 1][kWideDeps];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      rows[u] = rowOf(it, u);
    }
    if (FAST == 1) {
      uint32_t word[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int64_t v = vnext[u];
        candidate[u] = rows[u] < a.numRows && v >= a.ranges[0].min && v <= a.ranges[0].max;
        key[u] = static_cast<uint64_t>(v) - static_cast<uint64_t>(a.ranges[0].min) + 1;
      }
      // the window of presence words hangs on the wave's first key whether or not its row passes
      const bool firstInRange = candidate[0];
      if constexpr (RF > 0) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          candidate[u] = candidate[u] && fastFilterPass(a.rf, fnext[u]);
        }
      } else if constexpr (RF < 0) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          candidate[u] = candidate[u] && rowPasses(a.rf, rows[u]);
        }
      }
      if (mode == JMODE_ARRAY) {
        if (SPARSE && a.window) {
          // A wave of the listing probe covers 256 consecutive rows per iteration. When the probe
          // side is clustered by key (a fact table stored in key order) their presence bits sit in
          // a handful of neighbouring words: ONE coalesced load fetches the 64 words from the
          // first row's word on (2048 keys), every row takes its word from that window with a
          // lane permute, and only rows outside it (unordered probe sides) gather on their own.
          // A dependent gather costs the CU's address path ~80 cycles per wave instruction even
          // when all 64 lanes hit one line; the permute costs a tenth of that.
          const uint64_t w0 = readFirst64(firstInRange ? key[0] >> 5 : 0);
          const uint64_t mineWord = w0 + lane();
          const uint32_t win = a.present[mineWord < a.presentWords ? mineWord : a.presentWords - 1];
#pragma unroll
          for (int u = 0; u < kU; ++u) {
            const uint64_t idx = (key[u] >> 5) - w0;
            const uint32_t fromWindow = static_cast<uint32_t>(__shfl(static_cast<int>(win), static_cast<int>(idx & 63), kWave));
            word[u] = candidate[u] ? (idx < 64 ? fromWindow : a.present[key[u] >> 5]) : 0;
          }
        } else {
#pragma unroll
          for (int u = 0; u < kU; ++u) {
            word[u] = candidate[u] ? a.present[key[u] >> 5] : 0;
          }
        }
      }
      if (it + 1 < kIters) {
        const int64_t* kp = static_cast<const int64_t*>(a.keys[0].values);
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int64_t r = rowOf(it + 1, u);
          vnext[u] = VX355_KEY_LOAD(kp + (r < a.numRows ? r : a.numRows - 1));
          if constexpr (RF > 0) {
            fnext[u] = fastFilterLoad<RF>(a.rf, r < a.numRows ? r : a.numRows - 1);
          }
        }
      }
      if (mode == JMODE_ARRAY) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          candidate[u] = (word[u] >> (key[u] & 31)) & 1;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          hit[u] = candidate[u] ? a.head[key[u]] : kNoRow32;
        }
      } else if constexpr (WIDE > 0 && !SPARSE) {
        lookupFirstSlots<WIDE, kU>(a, key, candidate, hit, vals);
      } else {
        lookupFirstSlots<0, kU>(a, key, candidate, hit, nullptr);
      }
    } else if (fastKey) {
      const int64_t* kp = static_cast<const int64_t*>(a.keys[0].values);
      int64_t v[kU];
      int64_t src[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        src[u] = rows[u] < a.numRows ? rows[u] : a.numRows - 1;
      }
      if (fastKey == 2) {
        // dictionary-wrapped key (the probe input came through a FilterProject)
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          src[u] = a.keys[0].indices[src[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        v[u] = kp[src[u]];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        candidate[u] = rows[u] < a.numRows && v[u] >= a.ranges[0].min && v[u] <= a.ranges[0].max;
        key[u] = static_cast<uint64_t>(v[u]) - static_cast<uint64_t>(a.ranges[0].min) + 1;
      }
    } else if (mode != JMODE_HASH) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        candidate[u] = rows[u] < a.numRows && probeKey(a, rows[u], &key[u]);
      }
    }
    if constexpr (RF < 0) {
      if (FAST != 1 && mode != JMODE_HASH) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          candidate[u] = candidate[u] && rowPasses(a.rf, rows[u]);
        }
      }
    }
    if (FAST == 1) {
      // looked up above
    } else if (mode == JMODE_HASH) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        bool live = rows[u] < a.numRows;
        if constexpr (RF < 0) {
          live = live && rowPasses(a.rf, rows[u]);
        }
        hit[u] = live ? lookupGeneric(a, rows[u]) : kNoRow32;
      }
    } else if (mode == JMODE_ARRAY) {
      uint32_t word[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        word[u] = candidate[u] ? a.present[key[u] >> 5] : 0;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        candidate[u] = (word[u] >> (key[u] & 31)) & 1;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        hit[u] = candidate[u] ? a.head[key[u]] : kNoRow32;
      }
    } else {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        hit[u] = candidate[u] ? lookupSlots(a, key[u]) : kNoRow32;
      }
    }
    if constexpr (SPARSE) {
      // joins that list matches only, chains of one row, not null aware; 'mine' is the
      // wave's running hit count (uniform across the wave)
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const bool isMatch = rows[u] < a.numRows && hit[u] != kNoRow32;
        const uint64_t m = ballot(isMatch);
        if (isMatch) {
          if (a.probed) {
            a.probed[hit[u]] = 1;
          }
          const uint64_t at = mine + lanePrefix(m);
          if (at < kSparseWaveCap) {
            lds->list[threadIdx.x >> 6][at] =
                (static_cast<unsigned long long>(static_cast<uint32_t>(rows[u])) << 32) | hit[u];
          }
        }
        mine += popc64(m);
      }
      continue;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (rows[u] < a.numRows) {
        bool nullKey = false;
        if (a.nullAware && hit[u] == kNoRow32) {
          for (int k = 0; k < a.numKeys; ++k) {
            nullKey = nullKey || colIsNull(a.keys[k], rows[u]);
          }
          if (nullKey) {
            hit[u] = kNullKey32;
          }
        }
        // (written once, read by a later launch: past the caches, which hold table lines)
        __builtin_nontemporal_store(hit[u], a.hits + rows[u]);
        if constexpr (WIDE > 0 && !SPARSE) {
          if (isHit(hit[u])) {
            __builtin_nontemporal_store(vals[u][0], a.hitVals[0] + rows[u]);
            if (WIDE > 1) {
              __builtin_nontemporal_store(vals[u][1], a.hitVals[1] + rows[u]);
            }
          }
        }
        uint32_t matches = isHit(hit[u]) ? 1 : 0;
        if (matches && a.probed) {
          // setProbedFlag on every row of the chain (the rows listJoinResults hands out).
          for (uint32_t r = hit[u]; r != kNoRow32; r = a.next[r]) {
            a.probed[r] = 1;
          }
        }
        if (a.counts) {
          if (matches && listsMatches(a.joinType)) {
            uint32_t r = a.next[hit[u]];
            while (r != kNoRow32) {
              ++matches;
              r = a.next[r];
            }
          }
          a.counts[rows[u]] = outputCount(a.joinType, matches, nullKey);
        }
        mine += outputCount(a.joinType, matches, nullKey);
      }
    }
  }
  return mine;
}

template <int MODE, int FAST, bool SPARSE, int WIDE = 0, int RF = 0>
__global__ __launch_bounds__(256) void k_join_probe(ProbeArgs args) {
  __shared__ uint64_t waveSums[4];
  __shared__ __attribute__((aligned(16))) unsigned char sparseRaw[SPARSE ? sizeof(SparseLds) : 16];
  const ProbeArgs& a = args;
  SparseLds* lds = reinterpret_cast<SparseLds*>(sparseRaw);
  int parity = 0;
  for (int64_t tile = a.tileBegin + blockIdx.x; tile < a.numTiles; tile += gridDim.x, parity ^= 1) {
    uint64_t mine = probeTileBody<MODE, FAST, SPARSE, WIDE, RF>(a, tile, lds);
    if constexpr (SPARSE) {
      const int wave = threadIdx.x >> 6;
      if (lane() == 0) {
        lds->waveCount[parity][wave] = static_cast<uint32_t>(mine);
      }
      blockSync();
      uint32_t before = 0, total = 0;
      bool overflow = false;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t c = lds->waveCount[parity][w];
        before += w < wave ? c : 0;
        total += c;
        overflow = overflow || c > kSparseWaveCap;
      }
      if (!overflow) {  // uniform
        // the wave's own list, behind the lists of the waves that own earlier rows
        uint2* out = a.staged + tile * kSparseCap + before;
        for (uint32_t i = lane(); i < static_cast<uint32_t>(mine); i += 64) {
          const unsigned long long e = lds->list[wave][i];
          out[i] = make_uint2(static_cast<uint32_t>(e >> 32), static_cast<uint32_t>(e));
        }
        if (threadIdx.x == 0) {
          a.tileSums[tile] = total;
          a.tileDense[tile] = 0;
          if (total && a.countHits) {  // sample launch only: one address takes < 100 M atomics/s
            atomicAdd(reinterpret_cast<unsigned long long*>(a.sparseStats + 1), static_cast<unsigned long long>(total));
          }
        }
        continue;  // no second barrier: the next tile uses the other waveCount buffer, and a wave only
                   // overwrites its own list after every wave has passed this tile's barrier
      }
      // More hits than a staging segment holds: redo the tile in the dense form
      // (hits[] for every row); the output pass scans it like a dense tile.
      if (threadIdx.x == 0) {
        a.tileDense[tile] = 1;
        atomicAdd(reinterpret_cast<unsigned long long*>(a.sparseStats), 1ULL);
      }
      mine = probeTileBody<MODE, FAST, false, WIDE, RF>(a, tile, lds);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mine += shfl64(mine, lane() ^ off);
    }
    if (lane() == 0) {
      waveSums[threadIdx.x >> 6] = mine;
    }
    blockSync();
    if (threadIdx.x == 0) {
      a.tileSums[tile] = waveSums[0] + waveSums[1] + waveSums[2] + waveSums[3];
    }
    blockSync();
  }
}

// ---- range-partitioned probe (array mode, probe keys in no particular order) -----------------
// tools/gather_bench.hip: dependent random reads of a structure larger than the per-XCD L2
// retire at ~50 G/s on this chip whatever the occupancy (75 MB presence bitmap, probe keys in
// random order: 6 ms per 323 M probes); even an L2-resident structure stops at ~150 G/s
// (address processing in the CU's texture path). LDS does not have that limit. So when the
// probe keys of a batch are scattered (k_key_locality) and the join lists matches only, the
// probe side is range-partitioned: {key offset, probe row} records go to bin = key >> 20
// (one LDS-histogram pass, one scatter pass, as the radix aggregation does), then one
// workgroup per bin loads its 128 KB slice of the bitmap into LDS and streams its records
// against it. Hits leave as {probe row, build row} words and are put back into probe-row
// order by one radix sort of the HITS (few: the path is taken at hit rates <= 12.5 %).
constexpr int kPartShift = 20;                       // keys per bin = bits of the bitmap slice
constexpr int kPartSliceWords = 1 << (kPartShift - 5);  // u32 words of one slice: 128 KB
constexpr int kPartMaxBins = 4096;
constexpr int kPartTileRows = 32768;

__device__ inline bool ppRowPasses(const RowFilter& f, int64_t row) {
  if (f.numTerms == 0) {
    return true;
  }
  if (f.fast == 4) {
    return fastFilterPass(f, fastFilterLoad<4>(f, row));
  }
  return f.fast ? fastFilterPass(f, fastFilterLoad<8>(f, row)) : rowPasses(f, row);
}

struct PartArgs {
  const int64_t* keys;   // flat BIGINT probe keys
  int64_t numRows;
  int64_t numTiles;
  int64_t keyMin, keyMax;
  int32_t numBins;
  int32_t pad;
  uint32_t* hist;        // [bin][tile]
  const uint64_t* offsets;
  uint64_t* recs;        // {key offset inside the bin : 32 | probe row : 32}
  RowFilter rf;          // fused input filter: rows that fail are not scattered
};

__global__ __launch_bounds__(1024) void k_pp_count(PartArgs a) {
  __shared__ uint32_t hist[kPartMaxBins];
  for (int64_t tile = blockIdx.x; tile < a.numTiles; tile += gridDim.x) {
    for (int i = threadIdx.x; i < a.numBins; i += blockDim.x) {
      hist[i] = 0;
    }
    blockSync();
    const int64_t begin = tile * kPartTileRows;
    const int64_t end = begin + kPartTileRows < a.numRows ? begin + kPartTileRows : a.numRows;
    for (int64_t base = begin; base < end; base += 8 * 1024) {
      int64_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t r = base + u * 1024 + threadIdx.x;
        v[u] = r < end ? a.keys[r] : INT64_MIN;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t r = base + u * 1024 + threadIdx.x;
        if (r < end && v[u] >= a.keyMin && v[u] <= a.keyMax && ppRowPasses(a.rf, r)) {
          const uint64_t key = static_cast<uint64_t>(v[u]) - static_cast<uint64_t>(a.keyMin) + 1;
          atomicAdd(&hist[key >> kPartShift], 1u);
        }
      }
    }
    blockSync();
    for (int i = threadIdx.x; i < a.numBins; i += blockDim.x) {
      a.hist[static_cast<int64_t>(i) * a.numTiles + tile] = hist[i];
    }
    blockSync();
  }
}

// Scatter with an LDS sort in front: a tile (32 K rows, the unit of the global histogram) is
// handled in sub-tiles of 8192 records. Each sub-tile is counting-sorted by bin inside LDS
// (histogram, scan, placement), then written out: the records of one bin form a run of
// consecutive lanes storing to consecutive addresses, instead of 1024 lanes storing 8 bytes
// to 1024 unrelated lines (the plain scatter reached 1.6 TB/s with ~1100 open bins).
constexpr int kPartSub = 8192;

__global__ __launch_bounds__(1024) void k_pp_scatter(PartArgs a) {
  __shared__ unsigned long long binBase[kPartMaxBins];  // next free record of the bin inside this tile's range
  __shared__ uint32_t cnt[kPartMaxBins];                // sub-tile histogram, then placement cursor
  __shared__ uint32_t start[kPartMaxBins];              // sub-tile exclusive scan
  __shared__ uint64_t recs[kPartSub];
  __shared__ uint16_t binOf[kPartSub];
  __shared__ uint32_t waveTotals[16];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  for (int64_t tile = blockIdx.x; tile < a.numTiles; tile += gridDim.x) {
    for (int i = tid; i < a.numBins; i += blockDim.x) {
      binBase[i] = a.offsets[static_cast<int64_t>(i) * a.numTiles + tile];
    }
    const int64_t begin = tile * kPartTileRows;
    const int64_t end = begin + kPartTileRows < a.numRows ? begin + kPartTileRows : a.numRows;
    for (int64_t base = begin; base < end; base += kPartSub) {
      for (int i = tid; i < a.numBins; i += blockDim.x) {
        cnt[i] = 0;
      }
      blockSync();
      // 1. load + histogram
      uint64_t rec[8];
      uint32_t bin[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t r = base + u * 1024 + tid;
        const int64_t v = r < end ? a.keys[r] : INT64_MIN;
        bin[u] = 0xffffffffu;
        if (r < end && v >= a.keyMin && v <= a.keyMax && ppRowPasses(a.rf, r)) {
          const uint64_t key = static_cast<uint64_t>(v) - static_cast<uint64_t>(a.keyMin) + 1;
          bin[u] = static_cast<uint32_t>(key >> kPartShift);
          rec[u] = ((key & ((1ULL << kPartShift) - 1)) << 32) | static_cast<uint32_t>(r);
          atomicAdd(&cnt[bin[u]], 1u);
        }
      }
      blockSync();
      // 2. exclusive scan of the histogram (up to 4 bins per thread)
      uint32_t mine[4];
      uint32_t sum = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int b = tid * 4 + k;
        mine[k] = b < a.numBins ? cnt[b] : 0;
        sum += mine[k];
      }
      uint32_t incl = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, kWave);
        if (lane() >= off) {
          incl += o;
        }
      }
      if (lane() == 63) {
        waveTotals[wave] = incl;
      }
      blockSync();
      uint32_t run = incl - sum;
      for (int w = 0; w < wave; ++w) {
        run += waveTotals[w];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int b = tid * 4 + k;
        if (b < a.numBins) {
          start[b] = run;
          cnt[b] = run;  // placement cursor
        }
        run += mine[k];
      }
      blockSync();
      // 3. placement inside LDS
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (bin[u] != 0xffffffffu) {
          const uint32_t pos = atomicAdd(&cnt[bin[u]], 1u);
          recs[pos] = rec[u];
          binOf[pos] = static_cast<uint16_t>(bin[u]);
        }
      }
      blockSync();
      // 4. runs out: record i of the sorted sub-tile goes behind the bin's earlier records
      uint32_t total = 0;
      for (int w = 0; w < 16; ++w) {
        total += waveTotals[w];
      }
      for (uint32_t i = tid; i < total; i += 1024) {
        const uint32_t b = binOf[i];
        a.recs[binBase[b] + (i - start[b])] = recs[i];
      }
      blockSync();
      for (int b = tid; b < a.numBins; b += blockDim.x) {
        binBase[b] += cnt[b] - start[b];
      }
      blockSync();
    }
  }
}

// Count-free variant for <= kPartFastBins bins: sub-tiles of 16384 records (128 KB of LDS: the
// records carry their bin in bits 52..63, so no bin array is staged), runs twice as long as
// k_pp_scatter's; and no histogram pass in front - every bin owns a region of 1.5 x the even
// share of the batch + 4096 records, a sub-tile claims the space of each run with one atomic on
// the bin's cursor. Probe keys spread evenly over the build side's key range (what makes a probe
// side "scattered" in the first place) never fill a region; if one does fill (skewed keys), the
// overflow flag sends the batch through the counted passes instead.
constexpr int kPartFastBins = 1024;
constexpr int kPartFastSub = 16384;

struct PartFastArgs {
  const int64_t* keys;
  int64_t numRows;
  int64_t keyMin, keyMax;
  int32_t numBins;
  int32_t pad;
  uint64_t binCap;          // records per bin region
  uint32_t* binCount;       // cursors, zero on entry
  uint32_t* overflow;
  uint64_t* recs;           // bin b: recs[b * binCap ...]
  RowFilter rf;             // fused input filter: rows that fail are not scattered
};

__global__ __launch_bounds__(1024) void k_pp_scatter_fast(PartFastArgs a) {
  __shared__ unsigned long long binBase[kPartFastBins];
  __shared__ uint32_t cnt[kPartFastBins];
  __shared__ uint32_t start[kPartFastBins];
  __shared__ uint64_t recs[kPartFastSub];
  __shared__ uint32_t waveTotals[16];
  constexpr int R = kPartFastSub / 1024;
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  for (int i = tid; i < kPartFastBins; i += 1024) {
    cnt[i] = 0;
  }
  blockSync();
  const int64_t numSub = (a.numRows + kPartFastSub - 1) / kPartFastSub;
  // The keys of a sub-tile are loaded one sub-tile ahead: behind the placement into LDS and in front
  // of the write-out, so that their HBM latency overlaps the stores of the previous sub-tile (one
  // workgroup per CU: nothing else would). Rows past the end are clamped, then ignored.
  int64_t vnext[R];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kPartFastSub + u * 1024 + tid;
    vnext[u] = a.keys[r < a.numRows ? r : a.numRows - 1];
  }
  for (int64_t sub = blockIdx.x; sub < numSub; sub += gridDim.x) {
    const int64_t base = sub * kPartFastSub;
    uint64_t rec[R];
    uint32_t bin[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int64_t r = base + u * 1024 + tid;
      const int64_t v = r < a.numRows ? vnext[u] : INT64_MIN;
      bin[u] = 0xffffffffu;
      if (r < a.numRows && v >= a.keyMin && v <= a.keyMax && ppRowPasses(a.rf, r)) {
        const uint64_t key = static_cast<uint64_t>(v) - static_cast<uint64_t>(a.keyMin) + 1;
        bin[u] = static_cast<uint32_t>(key >> kPartShift);
        rec[u] = (static_cast<uint64_t>(bin[u]) << 52) | ((key & ((1ULL << kPartShift) - 1)) << 32) | static_cast<uint32_t>(r);
        atomicAdd(&cnt[bin[u]], 1u);
      }
    }
    blockSync();
    const uint32_t mine = tid < a.numBins ? cnt[tid] : 0;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(incl, off, kWave);
      if (lane() >= off) {
        incl += o;
      }
    }
    if (lane() == 63) {
      waveTotals[wave] = incl;
    }
    blockSync();
    uint32_t run = incl - mine;
    for (int w = 0; w < wave; ++w) {
      run += waveTotals[w];
    }
    if (tid < a.numBins) {
      start[tid] = run;
      cnt[tid] = run;
      if (mine != 0) {
        const uint32_t at = atomicAdd(&a.binCount[tid], mine);
        if (at + mine > a.binCap) {
          *a.overflow = 1;
          binBase[tid] = ~0ULL;
        } else {
          binBase[tid] = static_cast<uint64_t>(tid) * a.binCap + at;
        }
      }
    }
    blockSync();
#pragma unroll
    for (int u = 0; u < R; ++u) {
      if (bin[u] != 0xffffffffu) {
        recs[atomicAdd(&cnt[bin[u]], 1u)] = rec[u];
      }
    }
    blockSync();
    if (sub + gridDim.x < numSub) {
      const int64_t nextBase = (sub + gridDim.x) * kPartFastSub;
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int64_t r = nextBase + u * 1024 + tid;
        vnext[u] = a.keys[r < a.numRows ? r : a.numRows - 1];
      }
    }
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      total += waveTotals[w];
    }
    for (uint32_t i = tid; i < total; i += 1024) {
      const uint64_t rw = recs[i];
      const uint32_t b = static_cast<uint32_t>(rw >> 52);
      if (binBase[b] != ~0ULL) {
        a.recs[binBase[b] + (i - start[b])] = rw;
      }
    }
    blockSync();
    if (tid < a.numBins) {
      cnt[tid] = 0;
    }
    blockSync();
  }
}

struct PartProbeArgs {
  const uint64_t* recs;
  const uint32_t* binCount;  // count-free layout (k_pp_scatter_fast): bin b = binCount[b] records from b * binCap
  uint64_t binCap;
  const uint64_t* offsets;   // record offsets per (bin, tile) cell, bin major
  int64_t numTiles;
  int32_t numBins;
  int32_t pad;
  const uint32_t* present;
  uint64_t presentWords;     // u32 words of the whole bitmap
  const uint32_t* head;
  uint8_t* probed;
  uint64_t* pairs;           // out: {probe row : 32 | build row : 32}
  unsigned long long* numPairs;
};

constexpr int kPartWaveBuf = 128;  // hits a wave collects in LDS before it claims output space

// Writes the wave's collected hits behind one claim on the global cursor (a single HBM
// address takes < 100 M atomics/s: one atomic per HIT would cost more than the probe).
__device__ inline void ppFlushWave(const PartProbeArgs& a, const uint64_t* buf, uint32_t count) {
  if (count == 0) {
    return;
  }
  unsigned long long base = 0;
  if (lane() == 0) {
    base = atomicAdd(a.numPairs, static_cast<unsigned long long>(count));
  }
  base = shfl64(base, 0);
  for (uint32_t i = lane(); i < count; i += 64) {
    a.pairs[base + i] = buf[i];
  }
}

__global__ __launch_bounds__(1024) void k_pp_probe(PartProbeArgs a) {
  __shared__ uint32_t slice[kPartSliceWords];
  __shared__ uint64_t waveBuf[16][kPartWaveBuf];
  uint64_t* buf = waveBuf[threadIdx.x >> 6];
  uint32_t count = 0;  // uniform across the wave
  for (int32_t bin = blockIdx.x; bin < a.numBins; bin += gridDim.x) {
    const uint64_t begin = a.binCount ? static_cast<uint64_t>(bin) * a.binCap : a.offsets[static_cast<int64_t>(bin) * a.numTiles];
    const uint64_t end = a.binCount ? begin + a.binCount[bin] : a.offsets[static_cast<int64_t>(bin + 1) * a.numTiles];
    if (begin == end) {
      continue;  // uniform
    }
    const uint64_t wordBase = static_cast<uint64_t>(bin) * kPartSliceWords;
    for (int i = threadIdx.x; i < kPartSliceWords; i += blockDim.x) {
      slice[i] = wordBase + i < a.presentWords ? a.present[wordBase + i] : 0;
    }
    blockSync();
    // kPer records per lane and round, the next round's loaded before this round's are tested: one
    // workgroup per CU (the slice takes 128 KB of LDS), so the records in flight are all that hides
    // the HBM latency - four per lane without the look-ahead kept 32 KB per CU in flight (2.5 TB/s)
    constexpr int kPer = 8;
    uint64_t recNext[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const uint64_t i = begin + u * 1024 + threadIdx.x;
      recNext[u] = a.recs[i < end ? i : end - 1];
    }
    for (uint64_t at = begin; at < end; at += kPer * 1024) {
      uint64_t rec[kPer];
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        rec[u] = recNext[u];
      }
      if (at + kPer * 1024 < end) {
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
          const uint64_t i = at + kPer * 1024 + u * 1024 + threadIdx.x;
          recNext[u] = a.recs[i < end ? i : end - 1];
        }
      }
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        const uint64_t i = at + u * 1024 + threadIdx.x;
        const uint32_t off = static_cast<uint32_t>(rec[u] >> 32) & ((1u << kPartShift) - 1);  // bits 52.. may carry the bin
        const bool hit = i < end && ((slice[off >> 5] >> (off & 31)) & 1);
        const uint64_t m = ballot(hit);
        if (m == 0) {
          continue;
        }
        const uint32_t hits = static_cast<uint32_t>(popc64(m));
        if (count + hits > kPartWaveBuf) {
          ppFlushWave(a, buf, count);
          count = 0;
        }
        if (hit) {
          const uint32_t build = a.head[(static_cast<uint64_t>(bin) << kPartShift) + off];
          if (a.probed) {
            a.probed[build] = 1;
          }
          buf[count + lanePrefix(m)] = (static_cast<uint64_t>(static_cast<uint32_t>(rec[u])) << 32) | build;
        }
        count += hits;
      }
    }
    blockSync();  // the next bin overwrites the slice
  }
  ppFlushWave(a, buf, count);
}

// Average number of distinct 128-byte lines of the presence bitmap that the 64 keys of a wave
// touch, over 64 sample waves spread across the batch: ~1-4 when the probe side is clustered
// by key (a fact table stored in key order), ~64 when it is not.
__global__ __launch_bounds__(64) void k_key_locality(const int64_t* keys, int64_t numRows, uint32_t* out) {
  const int64_t waves = numRows / 64;
  if (waves == 0) {
    return;
  }
  const int64_t w = (waves * blockIdx.x) / gridDim.x;
  const uint64_t line = static_cast<uint64_t>(keys[w * 64 + lane()]) >> 10;
  bool first = true;
  for (int l = 0; l < 64; ++l) {
    const uint64_t other = shfl64(line, l);
    if (l < static_cast<int>(lane()) && other == line) {
      first = false;
    }
  }
  const uint64_t m = ballot(first);
  if (lane() == 0) {
    atomicAdd(out, static_cast<uint32_t>(popc64(m)));
  }
}

// ---- regrouped probe input (normalized-key tables beyond every cache) -------------------------
// A probe into a table of hundreds of MB is bound by the fabric's dependent-random-read rate
// (~33 G probes/s on config 5's shape, profiles/r05_bench_c5_one_gpu.json), not by HBM bytes. The
// reference answers the same problem on the CPU by partitioning the BUILD (parallelJoinBuild,
// exec/HashTable.cpp:1003-1203: thread i owns a contiguous range of buckets) and keeping 64
// probes in flight (:697-725). Here the slot array already is partitioned - the top bits of a
// slot number name a contiguous slice of the array - so the PROBE side is regrouped instead:
// vx355_join_probe_add_input_regrouped moves the rows of the batch (all columns) so that rows
// whose keys start their walk in the same slice are adjacent, and probes the regrouped batch slice
// by slice with the workgroups of one slice on one XCD: the slice (<= ~2 MiB) stays in that XCD's
// L2 while its rows stream past (tools/partjoin_parts.hip: 101 G probes/s). The mapping the
// operator emits refers to the regrouped batch, ascending, so there is no way back to pay for.
//   1. k_grp_hist     per tile of 32 K rows: rows per slice                      (8 B/row read)
//   2. scanU32ToU64   bin-major exclusive scan: every (slice, tile) owns a range of the output
//   3. k_grp_scatter  per sub-tile of 8 K rows: counting sort by slice in LDS, then column by
//                     column through LDS - coalesced reads, runs of a slice's rows written
//                     to consecutive addresses                               (2 x row bytes)
//   4. k_grp_tiles    probe tiles listed per XCD in slice order
//   5. k_join_probe_grouped  persistent workgroups, block b works for XCD b % 8
// Rows inside a slice keep their tile order; inside a sub-tile their order is whatever the LDS
// cursor hands out (the reference's exchange does not define a row order either).
constexpr int kGrpMaxBins = 4096;
constexpr int kGrpTileRows = 32768;
constexpr int kGrpSub = 8192;
constexpr int kGrpMaxCols = 16;

struct GroupArgs {
  ColView keys[kMaxKeys];
  KeyRange ranges[kMaxKeys];
  int32_t numKeys;
  int32_t nullAsValue;
  int32_t fastKey;      // one flat BIGINT key without nulls
  int32_t binShift;     // slice of a key = (twangMix64(key) & mask) >> binShift
  uint64_t mask;        // capacity - 1
  int32_t numBins;
  int32_t numCols;
  int64_t numRows;
  int64_t numTiles;     // of kGrpTileRows rows
  uint32_t* hist;            // [bin][tile]
  const uint64_t* offsets;   // exclusive scan of hist (bin major), offsets[numBins * numTiles] = numRows
  // the columns as 'parts' of at most 8 bytes (a 16-byte column is two): element r of part q sits at
  // in[q] + r * stride[q] and is width[q] bytes wide
  int32_t numParts;
  int32_t pad;
  const char* in[2 * kGrpMaxCols];   // nullptr: the part's value is the row number itself
  char* out[2 * kGrpMaxCols];
  int32_t stride[2 * kGrpMaxCols];
  int32_t outStride[2 * kGrpMaxCols];  // (differs from stride when columns are gathered into records)
  int32_t width[2 * kGrpMaxCols];
};

// The slice a row is probed in. Rows that are proven misses before any slot is read (a null key,
// a value outside the build side's range: lookupValueIds, VectorHasher.cpp:550-565) touch no
// slice; they are spread over the bins by their position.
__device__ inline uint32_t groupBinOf(const GroupArgs& a, int64_t row) {
  uint64_t key = 0;
  bool ok = true;
  if (a.fastKey) {
    const int64_t v = static_cast<const int64_t*>(a.keys[0].values)[row];
    ok = v >= a.ranges[0].min && v <= a.ranges[0].max;
    key = static_cast<uint64_t>(v) - static_cast<uint64_t>(a.ranges[0].min) + 1;
  } else {
    for (int k = 0; k < a.numKeys && ok; ++k) {
      const ColView& c = a.keys[k];
      if (colIsNull(c, row)) {
        ok = a.nullAsValue != 0;  // value id 0
        continue;
      }
      int64_t v;
      bool mappable;
      const uint64_t id = valueIdAt(c, colIndex(c, row), a.ranges[k], &v, &mappable);
      ok = id != 0;
      key += a.ranges[k].multiplier * id;
    }
  }
  return ok ? static_cast<uint32_t>((twangMix64(key) & a.mask) >> a.binShift)
            : static_cast<uint32_t>(row >> 13) & static_cast<uint32_t>(a.numBins - 1);
}

__global__ __launch_bounds__(1024) void k_grp_hist(GroupArgs a) {
  __shared__ uint32_t hist[kGrpMaxBins];
  for (int64_t tile = blockIdx.x; tile < a.numTiles; tile += gridDim.x) {
    for (int i = threadIdx.x; i < a.numBins; i += blockDim.x) {
      hist[i] = 0;
    }
    blockSync();
    const int64_t begin = tile * kGrpTileRows;
    const int64_t end = begin + kGrpTileRows < a.numRows ? begin + kGrpTileRows : a.numRows;
    for (int64_t base = begin; base < end; base += 8 * 1024) {
      uint32_t bin[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t r = base + u * 1024 + threadIdx.x;
        bin[u] = groupBinOf(a, r < end ? r : end - 1);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (base + u * 1024 + threadIdx.x < end) {
          atomicAdd(&hist[bin[u]], 1u);
        }
      }
    }
    blockSync();
    for (int i = threadIdx.x; i < a.numBins; i += blockDim.x) {
      a.hist[static_cast<int64_t>(i) * a.numTiles + tile] = hist[i];
    }
    blockSync();
  }
}

// Element 'r' of part q, zero extended.
__device__ inline uint64_t groupLoadPart(const GroupArgs& a, int q, int64_t r) {
  if (a.in[q] == nullptr) {
    return static_cast<uint64_t>(r);
  }
  const char* at = a.in[q] + r * a.stride[q];
  switch (a.width[q]) {
    case 8:
      return __builtin_nontemporal_load(reinterpret_cast<const uint64_t*>(at));
    case 4:
      return __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(at));
    case 2:
      return *reinterpret_cast<const uint16_t*>(at);
    default:
      return *reinterpret_cast<const uint8_t*>(at);
  }
}

__device__ inline void groupStorePart(const GroupArgs& a, int q, uint64_t row, uint64_t v) {
  char* at = a.out[q] + row * a.outStride[q];
  switch (a.width[q]) {
    case 8:
      *reinterpret_cast<uint64_t*>(at) = v;
      break;
    case 4:
      *reinterpret_cast<uint32_t*>(at) = static_cast<uint32_t>(v);
      break;
    case 2:
      *reinterpret_cast<uint16_t*>(at) = static_cast<uint16_t>(v);
      break;
    default:
      *reinterpret_cast<uint8_t*>(at) = static_cast<uint8_t>(v);
      break;
  }
}

// LDS (dynamic, sized by the host from the number of bins: groupScatterLds): the sorted sub-tile goes
// through a stage of HALF its rows at a time - 64 KB in all for up to 1024 bins, so that two
// workgroups share a CU and one's loads overlap the other's LDS phases.
template <bool HALF>
__host__ __device__ inline size_t groupScatterLds(int numBins) {
  return static_cast<size_t>(numBins) * 16 + kGrpSub * 2 + (HALF ? kGrpSub / 2 : kGrpSub) * 8;
}

template <bool HALF>
__global__ __launch_bounds__(1024) void k_grp_scatter(GroupArgs a) {
  constexpr int kGrpStage = HALF ? kGrpSub / 2 : kGrpSub;
  extern __shared__ __attribute__((aligned(16))) unsigned char grpLds[];
  unsigned long long* binBase = reinterpret_cast<unsigned long long*>(grpLds);  // next free output row of the bin inside this tile's range
  uint64_t* stage = reinterpret_cast<uint64_t*>(binBase + a.numBins);           // half of one part of the sub-tile, sorted
  uint32_t* cnt = reinterpret_cast<uint32_t*>(stage + kGrpStage);               // sub-tile histogram, then placement cursor
  uint32_t* start = cnt + a.numBins;                                            // sub-tile exclusive scan
  uint16_t* binAt = reinterpret_cast<uint16_t*>(start + a.numBins);             // bin of the i-th row of the sorted sub-tile
  __shared__ uint32_t waveTotals[16];
  constexpr int R = kGrpSub / 1024;
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  for (int64_t tile = blockIdx.x; tile < a.numTiles; tile += gridDim.x) {
    for (int i = tid; i < a.numBins; i += 1024) {
      binBase[i] = a.offsets[static_cast<int64_t>(i) * a.numTiles + tile];
    }
    const int64_t tileBegin = tile * kGrpTileRows;
    const int64_t end = tileBegin + kGrpTileRows < a.numRows ? tileBegin + kGrpTileRows : a.numRows;
    for (int64_t base = tileBegin; base < end; base += kGrpSub) {
      // the first part's values are on their way while the bins are computed; every later part is
      // loaded while the one before it goes through LDS
      uint64_t va[R], vb[R];
      auto load = [&](int q, uint64_t (&v)[R]) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
          const int64_t r = base + u * 1024 + tid;
          v[u] = groupLoadPart(a, q, r < end ? r : end - 1);
        }
      };
      load(0, va);
      for (int b = tid; b < a.numBins; b += 1024) {
        cnt[b] = 0;
      }
      blockSync();
      uint32_t bin[R];
      if (a.fastKey) {
        // part 0 IS the key column (the host lists it first): no second read of the keys
#pragma unroll
        for (int u = 0; u < R; ++u) {
          const int64_t r = base + u * 1024 + tid;
          const int64_t v = static_cast<int64_t>(va[u]);
          const bool ok = v >= a.ranges[0].min && v <= a.ranges[0].max;
          const uint64_t key = static_cast<uint64_t>(v) - static_cast<uint64_t>(a.ranges[0].min) + 1;
          bin[u] = ok ? static_cast<uint32_t>((twangMix64(key) & a.mask) >> a.binShift)
                      : static_cast<uint32_t>((r < end ? r : end - 1) >> 13) & static_cast<uint32_t>(a.numBins - 1);
        }
      } else {
#pragma unroll
        for (int u = 0; u < R; ++u) {
          const int64_t r = base + u * 1024 + tid;
          bin[u] = groupBinOf(a, r < end ? r : end - 1);
        }
      }
#pragma unroll
      for (int u = 0; u < R; ++u) {
        if (base + u * 1024 + tid < end) {
          atomicAdd(&cnt[bin[u]], 1u);
        } else {
          bin[u] = 0xffffffffu;
        }
      }
      blockSync();
      // exclusive scan of the histogram: bins 4 t .. 4 t + 3 belong to thread t
      constexpr int kBinsPerThread = kGrpMaxBins / 1024;
      uint32_t mineBins[kBinsPerThread];
      uint32_t mine = 0;
#pragma unroll
      for (int k = 0; k < kBinsPerThread; ++k) {
        const int b = tid * kBinsPerThread + k;
        mineBins[k] = b < a.numBins ? cnt[b] : 0;
        mine += mineBins[k];
      }
      uint32_t incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, kWave);
        if (lane() >= off) {
          incl += o;
        }
      }
      if (lane() == 63) {
        waveTotals[wave] = incl;
      }
      blockSync();
      uint32_t run = incl - mine;
      for (int w = 0; w < wave; ++w) {
        run += waveTotals[w];
      }
#pragma unroll
      for (int k = 0; k < kBinsPerThread; ++k) {
        const int b = tid * kBinsPerThread + k;
        if (b < a.numBins) {
          start[b] = run;
          cnt[b] = run;
        }
        run += mineBins[k];
      }
      blockSync();
      uint32_t pos[R];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        pos[u] = 0;
        if (bin[u] != 0xffffffffu) {
          pos[u] = atomicAdd(&cnt[bin[u]], 1u);
          binAt[pos[u]] = static_cast<uint16_t>(bin[u]);
        }
      }
      blockSync();
      const uint32_t total = static_cast<uint32_t>(base + kGrpSub <= end ? kGrpSub : end - base);
      uint64_t dst[R];
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const uint32_t i = k * 1024 + tid;
        dst[k] = 0;
        if (i < total) {
          const uint32_t b = binAt[i];
          dst[k] = binBase[b] + (i - start[b]);
        }
      }
      auto move = [&](int q, const uint64_t (&v)[R]) {
        constexpr int kRounds = HALF ? 2 : 1;
#pragma unroll
        for (int half = 0; half < kRounds; ++half) {
#pragma unroll
          for (int u = 0; u < R; ++u) {
            if (bin[u] != 0xffffffffu && (!HALF || (pos[u] >= kGrpStage) == (half == 1))) {
              stage[pos[u] - half * kGrpStage] = v[u];
            }
          }
          blockSync();
#pragma unroll
          for (int k = 0; k < R / kRounds; ++k) {
            const uint32_t i = half * kGrpStage + k * 1024 + tid;
            if (i < total) {
              groupStorePart(a, q, dst[half * (R / kRounds) + k], stage[i - half * kGrpStage]);
            }
          }
          blockSync();
        }
      };
      for (int q = 0; q < a.numParts; q += 2) {
        if (q + 1 < a.numParts) {
          load(q + 1, vb);
        }
        move(q, va);
        if (q + 1 < a.numParts) {
          if (q + 2 < a.numParts) {
            load(q + 2, va);
          }
          move(q + 1, vb);
        }
      }
      for (int b = tid; b < a.numBins; b += 1024) {
        binBase[b] += cnt[b] - start[b];
      }
      blockSync();
    }
  }
}

// The probe units (kGrpUnitRows rows of the regrouped batch: a quarter of a tile, so that the rows
// in flight on an XCD can be held to about one slice's worth even when a slice has few rows)
// listed per XCD: unit u belongs to the slice its first row is in, slice s to XCD s % 8; list x
// holds the units of slices x, x + 8, ... in that order. xcdStart[0..8] = where each list begins.
constexpr int kGrpUnitRows = 2048;
static_assert(kTileRows % kGrpUnitRows == 0, "units do not straddle tiles");

// unitList entry: {unit, its slice, its rank among the units of the slice, units of the slice}
__global__ __launch_bounds__(1024) void k_grp_units(const uint64_t* offsets, int64_t histTiles, int32_t numBins,
                                                    int64_t numUnits, int32_t unitRows, int4* unitList, uint32_t* xcdStart) {
  __shared__ uint32_t first[kGrpMaxBins + 1];  // first unit whose first row is at or behind the bin's start
  __shared__ uint32_t before[kGrpMaxBins];     // units of earlier bins on the same XCD
  __shared__ uint32_t listStart[9];
  const int tid = threadIdx.x;
  for (int b = tid; b <= numBins; b += 1024) {
    const uint64_t rowsBefore = offsets[static_cast<int64_t>(b) * histTiles];
    first[b] = static_cast<uint32_t>((rowsBefore + unitRows - 1) / unitRows);
  }
  blockSync();
  if (tid < 8) {
    uint32_t run = 0;
    for (int b = tid; b < numBins; b += 8) {
      before[b] = run;
      run += first[b + 1] - first[b];
    }
    listStart[tid + 1] = run;
  }
  blockSync();
  if (tid == 0) {
    listStart[0] = 0;
    for (int x = 1; x <= 8; ++x) {
      listStart[x] += listStart[x - 1];
    }
    if (blockIdx.x == 0) {
      for (int x = 0; x <= 8; ++x) {
        xcdStart[x] = listStart[x];
      }
    }
  }
  blockSync();
  // every unit finds its bin (the last bin that begins at or before it) and from it its place
  const int64_t u = static_cast<int64_t>(blockIdx.x) * 1024 + tid;
  if (u < numUnits) {
    int lo = 0, hi = numBins;  // first[lo] <= u < first[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (first[mid] <= static_cast<uint32_t>(u)) {
        lo = mid;
      } else {
        hi = mid;
      }
    }
    // empty bins share their 'first' with the next bin: take the last bin with first[b] <= u, which owns u
    const uint32_t rank = static_cast<uint32_t>(u) - first[lo];
    unitList[listStart[lo & 7] + before[lo] + rank] =
        make_int4(static_cast<int32_t>(u), lo, static_cast<int32_t>(rank), static_cast<int32_t>(first[lo + 1] - first[lo]));
  }
}

// k_join_probe over a regrouped batch, one unit per workgroup: block b takes the (b / 8)-th unit of
// the list of XCD b % 8 (workgroups are dealt to the XCDs round robin and start in index order -
// observed, not promised: another placement costs speed, never correctness). The launch asks for
// LDS it does not use so that only a few workgroups fit a CU: the rows in flight on an XCD then
// span about one slice, which is what lets the slice live in that XCD's L2. Lists longer than
// gridDim.x / 8 (skewed slices) wrap around. tileSums[] is zero on entry.
// Before it probes, a workgroup streams its share of the slice (sliceBytes / units of the slice)
// through plain coalesced loads: the slice reaches the L2 in whole lines at streaming speed instead
// of one dependent miss per first touch of a line (with few probe rows per slice - chunked input -
// those misses were 19 % of the accesses, profiles/r06_c5_counters.md).
template <int FAST, int WIDE, int UNIT>
__global__ __launch_bounds__(256) void k_join_probe_grouped(ProbeArgs args, const int4* unitList, const uint32_t* xcdStart,
                                                             const uint4* tableBase, uint64_t sliceUnits16) {
  __shared__ uint64_t waveSums[4];
  const ProbeArgs& a = args;
  const uint32_t xcd = blockIdx.x & 7;
  const uint32_t perXcd = gridDim.x >> 3;
  const uint32_t end = xcdStart[xcd + 1];
  uint32_t touched = 0;
  for (uint32_t j = xcdStart[xcd] + (blockIdx.x >> 3); j < end; j += perXcd) {
    const int4 info = unitList[j];
    const int64_t unit = info.x;
    if (sliceUnits16) {
      const uint4* slice = tableBase + static_cast<uint64_t>(info.y) * sliceUnits16;
      const uint64_t lo = sliceUnits16 * static_cast<uint64_t>(info.z) / static_cast<uint64_t>(info.w);
      const uint64_t hi = sliceUnits16 * (static_cast<uint64_t>(info.z) + 1) / static_cast<uint64_t>(info.w);
#pragma unroll 4
      for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) {
        touched ^= slice[i].x;
      }
    }
    uint64_t mine = probeTileBody<JMODE_NORMALIZED, FAST, false, WIDE, 0, 256, UNIT>(a, unit, nullptr);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mine += shfl64(mine, lane() ^ off);
    }
    if (lane() == 0) {
      waveSums[threadIdx.x >> 6] = mine;
    }
    blockSync();
    if (threadIdx.x == 0) {
      const uint64_t total = waveSums[0] + waveSums[1] + waveSums[2] + waveSums[3];
      if (total) {
        atomicAdd(reinterpret_cast<unsigned long long*>(a.tileSums + unit / (kTileRows / UNIT)),
                  static_cast<unsigned long long>(total));
      }
      if (UNIT == kTileRows / 4 && a.unitSums) {
        a.unitSums[unit] = static_cast<uint32_t>(total);
      }
    }
    blockSync();
  }
  if (touched == 0x9e3779b9u && a.numRows < 0) {
    a.hits[0] = touched;  // never: keeps the streaming loads alive
  }
}

__global__ __launch_bounds__(256) void k_emit_pairs(const uint64_t* sorted, int64_t begin, int32_t n, int32_t wantBuild,
                                                     int32_t* mapping, int32_t* buildRows) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint64_t w = sorted[begin + i];
    mapping[i] = static_cast<int32_t>(w >> 32);
    if (buildRows) {
      buildRows[i] = wantBuild ? static_cast<int32_t>(static_cast<uint32_t>(w)) : -1;
    }
  }
}

// ---- normalized-key table assembled group by group in LDS ---------------------------------------
// Inserting rows one by one into a table of hundreds of MB is a stream of random read-modify-writes:
// ~20 G/s per chip whatever their locality (profiles/r02_atomic_bench.txt), 130 bytes written to HBM
// per row (profiles/r06_c5_counters.md) - 2.7 ms for the 20 M rows of config 5's one-GPU shape. The
// reference parallelises the same step by giving every thread a contiguous range of buckets and the
// rows that hash there (parallelJoinBuild / buildJoinPartition, exec/HashTable.cpp:1003-1203,
// :1255). Here a range is a 'group' of slots that fits LDS (128 KB), a row's walk wraps inside its
// group (nextSlot), and:
//   1. k_norm_keys          the normalized key of every build row                 (16 B/row)
//   2. k_grp_hist/_scatter  {key, row, dependents} records grouped by the top bits of the slot
//                           number - the passes that regroup a probe batch      (~ 2 x 32 B/row)
//   3. k_lds_build          one workgroup per group: the records of the group's bin are streamed
//                           (L2: the workgroups of one bin follow each other on one XCD), those of
//                           the group inserted into the LDS image with LDS atomics, the image
//                           written out in whole lines - no fill pass, no HBM atomic.
struct LdsBuildArgs {
  const char* recs;           // records of recBytes: {u64 key, u32 row, u32 -, [u64 dep0, u64 dep1]}
  const uint64_t* offsets;    // bin-major scan of the regroup: bin b starts at offsets[b * histTiles]
  int64_t histTiles;
  int32_t groupsPerBinShift;  // group g belongs to bin g >> groupsPerBinShift
  int32_t slotShift;          // 4 / 5 (records have the size of a slot)
  int32_t groupShift;         // log2(slots per group)
  int32_t dropDups;
  uint64_t mask;              // capacity - 1
  uint64_t homeMask;
  char* slots;
  uint32_t* next;
  BuildCounters* counters;
  int32_t numGroups;
  int32_t pad;
};

__global__ __launch_bounds__(256) void k_norm_keys(InsertArgs a, uint64_t* out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; row < a.numRows; row += stride) {
    out[row] = buildKey(a, row);
  }
}

__global__ __launch_bounds__(1024) void k_lds_build(LdsBuildArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char image[];
  __shared__ uint32_t tallies[3];  // distinct, duplicates, overflow
  // XCD-aware: block b works for XCD b % 8 and takes the (b / 8)-th group of that XCD's contiguous
  // range of groups, so the groups of one bin (which stream the same records) share an L2
  const uint32_t perXcd = (static_cast<uint32_t>(a.numGroups) + 7) / 8;
  const uint32_t group = (blockIdx.x & 7) * perXcd + (blockIdx.x >> 3);
  if (group >= static_cast<uint32_t>(a.numGroups)) {
    return;
  }
  const uint32_t groupSlots = 1u << a.groupShift;
  const uint32_t units = groupSlots << (a.slotShift - 4);
  for (uint32_t i = threadIdx.x; i < units; i += 1024) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (a.slotShift == 4 || (i & 1) == 0) {
      v = make_uint4(0xffffffffu, 0xffffffffu, kNoRow32, 0);
    }
    reinterpret_cast<uint4*>(image)[i] = v;
  }
  if (threadIdx.x < 3) {
    tallies[threadIdx.x] = 0;
  }
  blockSync();
  const uint32_t bin = group >> a.groupsPerBinShift;
  const uint64_t begin = a.offsets[static_cast<int64_t>(bin) * a.histTiles];
  const uint64_t end = a.offsets[static_cast<int64_t>(bin + 1) * a.histTiles];
  const int recShift = a.slotShift;
  uint32_t distinct = 0, dups = 0;
  // (Tried and dropped, profiles/r06_c5_variants.md: the next round's records loaded before this round's
  // are inserted - 4.0 instead of 1.6 ms, the second set of registers costs the occupancy that hid the
  // latency; the fields as separate arrays, so that the scan reads 8 instead of 32 bytes per row - 2.1 ms,
  // three dependent loads per inserted row instead of one.)
  constexpr int kPer = 4;
  for (uint64_t base = begin; base < end; base += kPer * 1024) {
    uint4 head[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const uint64_t i = base + u * 1024 + threadIdx.x;
      head[u] = *reinterpret_cast<const uint4*>(a.recs + ((i < end ? i : end - 1) << recShift));
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const uint64_t i = base + u * 1024 + threadIdx.x;
      const uint64_t key = (static_cast<uint64_t>(head[u].y) << 32) | head[u].x;
      const uint64_t home = twangMix64(key) & a.homeMask;
      if (i >= end || (home >> a.groupShift) != group) {
        continue;
      }
      const uint32_t row = head[u].z;
      uint32_t pos = static_cast<uint32_t>(home) & (groupSlots - 1);
      bool placed = false;
      for (uint32_t probes = 0; probes < groupSlots; ++probes) {
        unsigned char* slot = image + (static_cast<size_t>(pos) << a.slotShift);
        const unsigned long long old =
            atomicCAS(reinterpret_cast<unsigned long long*>(slot), static_cast<unsigned long long>(kEmptyKey),
                      static_cast<unsigned long long>(key));
        if (old == kEmptyKey) {
          if (a.slotShift == 5) {
            // (the dependents share the 32-byte sector of the key just read)
            *reinterpret_cast<uint4*>(slot + 16) = *reinterpret_cast<const uint4*>(a.recs + (i << recShift) + 16);
          }
          ++distinct;
        }
        if (old == kEmptyKey || old == key) {
          if (old == key && a.dropDups) {
            a.next[row] = kNoRow32;  // the claiming row stays the key's only row
          } else {
            // pushNext (HashTable.cpp:1412-1418): the new row becomes the head of the key's chain
            a.next[row] = atomicExch(reinterpret_cast<uint32_t*>(slot + 8), row);
            dups += old == key ? 1 : 0;
          }
          placed = true;
          break;
        }
        pos = (pos + 1) & (groupSlots - 1);
      }
      if (!placed) {
        tallies[2] = 1;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    distinct += __shfl_xor(distinct, off, kWave);
    dups += __shfl_xor(dups, off, kWave);
  }
  if (lane() == 0) {
    atomicAdd(&tallies[0], distinct);
    atomicAdd(&tallies[1], dups);
  }
  blockSync();
  uint4* out = reinterpret_cast<uint4*>(a.slots + ((static_cast<size_t>(group) << a.groupShift) << a.slotShift));
  for (uint32_t i = threadIdx.x; i < units; i += 1024) {
    out[i] = reinterpret_cast<const uint4*>(image)[i];
  }
  if (threadIdx.x == 0) {
    if (tallies[0]) {
      atomicAdd(&a.counters->numDistinct, tallies[0]);
    }
    if (tallies[1]) {
      atomicAdd(&a.counters->duplicates, tallies[1]);
    }
    if (tallies[2]) {
      a.counters->tableFull = 1;
    }
  }
}

// ---- the join's extra filter (HashProbe::evalFilter, HashProbe.cpp:1713) -------------------
constexpr int kMaxJoinFilterTerms = 4;

struct JoinFilterTerm {
  ColView leftProbe, rightProbe;  // operand = probe column
  int32_t leftSide;               // 0 probe column, 1 build dependent
  int32_t leftDep, rightDep;      // operand = build dependent (index)
  int32_t cmp;
  int32_t rightKind;              // 0 constant, 1 probe column, 2 build dependent
  int32_t constKind;
  int64_t i64;
  double f64;
  uint4 str;                      // constant string as an inline StringView
};

struct JoinFilterArgs {
  int32_t numTerms;
  int32_t pad;
  JoinFilterTerm terms[kMaxJoinFilterTerms];
  const char* depVals[kMaxDeps];
  const uint8_t* depValid[kMaxDeps];
  int32_t depWidth[kMaxDeps];
  int32_t depKind[kMaxDeps];
};

struct FilterOperand {
  bool null;
  int cls;  // 0 int64, 1 double, 2 inline string
  int64_t i;
  double d;
  uint4 s;
};

__device__ inline FilterOperand probeOperand(const ColView& c, int64_t row) {
  FilterOperand o{};
  if (colIsNull(c, row)) {
    o.null = true;
    return o;
  }
  const int64_t i = colIndex(c, row);
  if (c.kind == VX355_VARCHAR || c.kind == VX355_VARBINARY) {
    o.cls = 2;
    o.s = static_cast<const uint4*>(c.values)[i];
  } else if (c.kind == VX355_REAL || c.kind == VX355_DOUBLE) {
    o.cls = 1;
    o.d = loadDouble(c, i);
  } else {
    o.i = loadInt64(c, i);
  }
  return o;
}

__device__ inline FilterOperand buildOperand(const JoinFilterArgs& f, int dep, uint32_t row) {
  FilterOperand o{};
  if (!f.depValid[dep][row]) {
    o.null = true;
    return o;
  }
  const int w = f.depWidth[dep];
  const char* p = f.depVals[dep] + static_cast<int64_t>(row) * (w == 0 ? 1 : w);
  switch (f.depKind[dep]) {
    case VX355_BOOLEAN:
    case VX355_TINYINT:
      o.i = *reinterpret_cast<const int8_t*>(p);
      break;
    case VX355_SMALLINT:
      o.i = *reinterpret_cast<const int16_t*>(p);
      break;
    case VX355_INTEGER:
      o.i = *reinterpret_cast<const int32_t*>(p);
      break;
    case VX355_BIGINT:
      o.i = *reinterpret_cast<const int64_t*>(p);
      break;
    case VX355_REAL:
      o.cls = 1;
      o.d = *reinterpret_cast<const float*>(p);
      break;
    case VX355_DOUBLE:
      o.cls = 1;
      o.d = *reinterpret_cast<const double*>(p);
      break;
    default:
      o.cls = 2;
      o.s = *reinterpret_cast<const uint4*>(p);
      break;
  }
  return o;
}


// True when every term holds for the pair; a null operand fails its term.
__device__ inline bool evalJoinFilter(const JoinFilterArgs& f, int64_t probeRow, uint32_t buildRow) {
  for (int t = 0; t < f.numTerms; ++t) {
    const JoinFilterTerm& term = f.terms[t];
    const FilterOperand l =
        term.leftSide == 0 ? probeOperand(term.leftProbe, probeRow) : buildOperand(f, term.leftDep, buildRow);
    FilterOperand r{};
    if (term.rightKind == 1) {
      r = probeOperand(term.rightProbe, probeRow);
    } else if (term.rightKind == 2) {
      r = buildOperand(f, term.rightDep, buildRow);
    } else if (term.constKind == VX355_BIGINT) {
      r.i = term.i64;
    } else if (term.constKind == VX355_DOUBLE) {
      r.cls = 1;
      r.d = term.f64;
    } else {
      r.cls = 2;
      r.s = term.str;
    }
    if (l.null || r.null) {
      return false;
    }
    bool ok;
    if (l.cls == 2 || r.cls == 2) {
      // inline strings: size, prefix and the 8 bytes behind it (unused bytes are zero)
      const bool eq = l.cls == r.cls &&
          stringImagesEqual(static_cast<uint64_t>(l.s.x) | (static_cast<uint64_t>(l.s.y) << 32),
                            static_cast<uint64_t>(l.s.z) | (static_cast<uint64_t>(l.s.w) << 32),
                            static_cast<uint64_t>(r.s.x) | (static_cast<uint64_t>(r.s.y) << 32),
                            static_cast<uint64_t>(r.s.z) | (static_cast<uint64_t>(r.s.w) << 32));
      ok = term.cmp == VX355_CMP_EQ ? eq : !eq;
    } else if (l.cls == 0 && r.cls == 0) {
      ok = compareValues<int64_t>(term.cmp, l.i, r.i);
    } else {
      ok = compareValues<double>(term.cmp, l.cls == 0 ? static_cast<double>(l.i) : l.d,
                                 r.cls == 0 ? static_cast<double>(r.i) : r.d);
    }
    if (!ok) {
      return false;
    }
  }
  return true;
}

struct FilterPassArgs {
  JoinFilterArgs f;
  uint32_t* hits;      // in: first row of the chain; out: first PASSING row, or kNoRow32
  uint32_t* counts;    // out: output rows of the probe row
  uint64_t* tileSums;
  const uint32_t* next;
  uint8_t* probed;     // right / full / right semi / right anti: flags of the passing pairs
  int64_t numRows;
  int64_t numTiles;
  int32_t joinType;
  int32_t pad;
};

// Second pass of a probe with an extra filter (or of any probe whose per-row output
// has to be recomputed from hits[]): walks every chain, evaluates the filter per
// pair, turns a probe row without a passing pair into a miss, counts the output
// rows and the tile sums, sets the probed flags of the passing pairs.
__global__ __launch_bounds__(256) void k_join_filter(FilterPassArgs a) {
  __shared__ uint64_t waveSums[4];
  const bool lists = listsMatches(a.joinType);
  for (int64_t tile = blockIdx.x; tile < a.numTiles; tile += gridDim.x) {
    uint64_t mine = 0;
    for (int it = 0; it < kTileRows / 256; ++it) {
      const int64_t r = tile * kTileRows + it * 256 + threadIdx.x;
      if (r >= a.numRows) {
        continue;
      }
      const uint32_t hit = a.hits[r];
      uint32_t passing = 0;
      uint32_t first = kNoRow32;
      if (isHit(hit)) {
        for (uint32_t b = hit; b != kNoRow32; b = a.next[b]) {
          if (a.f.numTerms == 0 || evalJoinFilter(a.f, r, b)) {
            if (first == kNoRow32) {
              first = b;
            }
            ++passing;
            if (a.probed) {
              a.probed[b] = 1;
            }
            if (!lists && !a.probed) {
              break;  // "does any pair pass" is all the semi / anti kinds ask
            }
          }
        }
        a.hits[r] = first;
      }
      const uint32_t c = outputCount(a.joinType, passing, hit == kNullKey32);
      a.counts[r] = c;
      mine += c;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mine += shfl64(mine, lane() ^ off);
    }
    if (lane() == 0) {
      waveSums[threadIdx.x >> 6] = mine;
    }
    blockSync();
    if (threadIdx.x == 0) {
      a.tileSums[tile] = waveSums[0] + waveSums[1] + waveSums[2] + waveSums[3];
    }
    blockSync();
  }
}

// ---- null-aware joins with an extra filter (HashProbe::evalFilterForNullAwareJoin,
// HashProbe.cpp:1639-1700) -----------------------------------------------------------------------
// x [NOT] IN (SELECT y FROM build WHERE filter(probe, build)) is three valued. After k_join_filter
// a probe row is TRUE when a build row with an equal key passes the filter. Of the others,
//   a row whose key is not null is NULL iff the filter passes on some build row whose key IS null,
//   a row whose key is null     is NULL iff the filter passes on ANY build row
// (the reference pairs them with listNullKeyRows / listAllRows), else FALSE. hits[r] becomes
// kNullKey32 for NULL and kNoRow32 for FALSE; NOT IN (anti) emits the FALSE rows, IN as a column
// (left semi project) reports first match / -2 / -1.
__global__ __launch_bounds__(256) void k_list_null_keys(const uint8_t* keyNull, int64_t n, uint32_t* rows,
                                                         uint32_t* count) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t rounds = (n + stride - 1) / stride;
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (int64_t r = 0; r < rounds; ++r, i += stride) {
    const bool isNull = i < n && keyNull[i] != 0;
    const uint64_t m = ballot(isNull);
    if (m == 0) {
      continue;
    }
    uint32_t base = 0;
    const int leader = __ffsll(static_cast<long long>(m)) - 1;
    if (lane() == leader) {
      base = atomicAdd(count, static_cast<uint32_t>(popc64(m)));
    }
    base = __shfl(base, leader, kWave);
    if (isNull) {
      rows[base + lanePrefix(m)] = static_cast<uint32_t>(i);
    }
  }
}

// Probe rows that k_na_resolve has to look at: no passing pair with an equal key.
__global__ __launch_bounds__(256) void k_na_pending(const uint32_t* hits, int64_t n, int64_t numNullKeyRows,
                                                     uint32_t* pending, uint32_t* count) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t rounds = (n + stride - 1) / stride;
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (int64_t r = 0; r < rounds; ++r, i += stride) {
    // a row with a non-null key only has candidates when the build side holds null keys
    const bool want = i < n && !isHit(hits[i]) && (hits[i] == kNullKey32 || numNullKeyRows > 0);
    const uint64_t m = ballot(want);
    if (m == 0) {
      continue;
    }
    uint32_t base = 0;
    const int leader = __ffsll(static_cast<long long>(m)) - 1;
    if (lane() == leader) {
      base = atomicAdd(count, static_cast<uint32_t>(popc64(m)));
    }
    base = __shfl(base, leader, kWave);
    if (want) {
      pending[base + lanePrefix(m)] = static_cast<uint32_t>(i);
    }
  }
}

struct NullAwareArgs {
  JoinFilterArgs f;
  uint32_t* hits;
  const uint32_t* pending;
  uint32_t numPending;
  uint32_t pad;
  const uint32_t* nullKeyRows;
  int64_t numNullKeyRows;
  int64_t numBuildRows;
};

// One wave per pending probe row: the lanes share out the candidate build rows.
__global__ __launch_bounds__(256) void k_na_resolve(NullAwareArgs a) {
  const uint32_t waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; p < a.numPending; p += waves) {
    const uint32_t row = a.pending[p];
    const bool nullKey = a.hits[row] == kNullKey32;
    const int64_t candidates = nullKey ? a.numBuildRows : a.numNullKeyRows;
    bool found = false;
    for (int64_t base = 0; base < candidates && !found; base += 64) {
      const int64_t c = base + lane();
      bool pass = false;
      if (c < candidates) {
        const uint32_t b = nullKey ? static_cast<uint32_t>(c) : a.nullKeyRows[c];
        pass = evalJoinFilter(a.f, row, b);
      }
      found = ballot(pass) != 0;
    }
    if (lane() == 0) {
      a.hits[row] = found ? kNullKey32 : kNoRow32;
    }
  }
}

// Output counts and tile sums from the resolved hits[] (TRUE = a build row, kNullKey32 = NULL,
// kNoRow32 = FALSE).
__global__ __launch_bounds__(256) void k_na_counts(const uint32_t* hits, int64_t numRows, int64_t numTiles,
                                                    int32_t joinType, uint32_t* counts, uint64_t* tileSums) {
  __shared__ uint64_t waveSums[4];
  for (int64_t tile = blockIdx.x; tile < numTiles; tile += gridDim.x) {
    uint64_t mine = 0;
    for (int it = 0; it < kTileRows / 256; ++it) {
      const int64_t r = tile * kTileRows + it * 256 + threadIdx.x;
      if (r < numRows) {
        const uint32_t c = joinType == VX355_JOIN_ANTI ? (hits[r] == kNoRow32 ? 1 : 0) : 1;
        counts[r] = c;
        mine += c;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mine += shfl64(mine, lane() ^ off);
    }
    if (lane() == 0) {
      waveSums[threadIdx.x >> 6] = mine;
    }
    blockSync();
    if (threadIdx.x == 0) {
      tileSums[tile] = waveSums[0] + waveSums[1] + waveSums[2] + waveSums[3];
    }
    blockSync();
  }
}

// ---- counting joins (INTERSECT ALL / EXCEPT ALL, HashProbe.cpp:1345-1365) -------------------
// remaining[head row] = occurrences of the key not yet consumed by a probe row.
// Probe rows of one batch that hit the same key are ranked in probe-row order
// (sorted {head, probe row} words); the first 'remaining' of them consume one
// occurrence each, exactly like the reference's row-at-a-time loop.
__global__ __launch_bounds__(256) void k_hit_bits(const uint32_t* hits, int64_t n, uint64_t* bits) {
  const int64_t numWords = (n + 63) >> 6;
  const int64_t waveStride = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t w = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; w < numWords;
       w += waveStride) {
    const int64_t r = (w << 6) + lane();
    const uint64_t word = ballot(r < n && isHit(hits[r]));
    if (lane() == 0) {
      bits[w] = word;
    }
  }
}

__global__ __launch_bounds__(256) void k_hit_words(const uint32_t* hits, const int32_t* rows, int64_t n,
                                                    uint64_t* words) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t row = static_cast<uint32_t>(rows[i]);
    words[i] = (static_cast<uint64_t>(hits[row]) << 32) | row;
  }
}

// Position of the first word of 'sorted' whose head is not smaller than 'head', in [0, hi].
__device__ inline int64_t lowerBoundHead(const uint64_t* sorted, int64_t hi, uint32_t head) {
  int64_t lo = 0;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (static_cast<uint32_t>(sorted[mid] >> 32) < head) {
      lo = mid + 1;
    } else {
      hi = mid;
    }
  }
  return lo;
}

// Counting joins: the hits of a batch sorted by {chain head, probe row}. The first remaining[head]
// rows of every run consume one occurrence each and stay hits, the rest become misses
// (HashProbe's counting semi / anti joins consume in probe-row order). Every lane finds its own
// rank in its run by binary search - a run of any length costs each lane log2(n) reads, so a
// skewed probe side (INTERSECT ALL over one hot key) does not serialise on one lane.
__global__ __launch_bounds__(256) void k_count_consume(const uint64_t* sorted, int64_t n, const uint32_t* remaining,
                                                        uint32_t* hits) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t head = static_cast<uint32_t>(sorted[i] >> 32);
    const bool first = i == 0 || static_cast<uint32_t>(sorted[i - 1] >> 32) != head;
    const int64_t rank = first ? 0 : i - lowerBoundHead(sorted, i, head);
    if (rank >= static_cast<int64_t>(remaining[head])) {
      hits[static_cast<uint32_t>(sorted[i])] = kNoRow32;
    }
  }
}

// Second pass (behind the launch boundary: k_count_consume reads 'remaining'): the lane at the
// start of a run takes what the run consumed off the head's remaining count.
__global__ __launch_bounds__(256) void k_count_settle(const uint64_t* sorted, int64_t n, uint32_t* remaining) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t head = static_cast<uint32_t>(sorted[i] >> 32);
    if (i > 0 && static_cast<uint32_t>(sorted[i - 1] >> 32) == head) {
      continue;
    }
    // end of the run: first word whose head is larger
    int64_t lo = i + 1, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (static_cast<uint32_t>(sorted[mid] >> 32) <= head) {
        lo = mid + 1;
      } else {
        hi = mid;
      }
    }
    const uint64_t length = static_cast<uint64_t>(lo - i);
    const uint32_t left = remaining[head];
    remaining[head] = length >= left ? 0 : left - static_cast<uint32_t>(length);
  }
}

// ---- listJoinResults -------------------------------------------------------------------
// Single-block exclusive scan; offsets has n + 1 entries (last = total). 4096
// cells per iteration: coalesced loads of four consecutive cells per lane, wave
// scans with shuffles, one LDS exchange of the 16 wave totals.
__global__ __launch_bounds__(1024) void k_scan_u64(const uint64_t* in, int64_t n, uint64_t* offsets) {
  __shared__ uint64_t waveTotal[16];
  __shared__ uint64_t carryShared;
  const int t = threadIdx.x;
  const int wave = t >> 6;
  if (t == 0) {
    carryShared = 0;
  }
  blockSync();
  for (int64_t base = 0; base < n; base += 4096) {
    const int64_t i0 = base + static_cast<int64_t>(t) * 4;
    uint64_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = i0 + k < n ? in[i0 + k] : 0;
    }
    const uint64_t mine = v[0] + v[1] + v[2] + v[3];
    uint64_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint64_t o = shfl64(incl, lane() >= off ? lane() - off : lane());
      if (lane() >= off) {
        incl += o;
      }
    }
    if (lane() == 63) {
      waveTotal[wave] = incl;
    }
    blockSync();
    uint64_t run = carryShared + (incl - mine);
    uint64_t total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const uint64_t wt = waveTotal[w];
      run += w < wave ? wt : 0;
      total += wt;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i0 + k < n) {
        offsets[i0 + k] = run;
      }
      run += v[k];
    }
    blockSync();
    if (t == 0) {
      carryShared += total;
    }
    blockSync();
  }
  if (t == 0) {
    offsets[n] = carryShared;
  }
}

struct EmitArgs {
  const uint32_t* hits;
  const uint32_t* counts;  // nullptr: derive from hits (no duplicate chains)
  const uint32_t* next;
  const uint64_t* tileOffsets;
  int64_t numRows;
  int64_t firstTile;
  uint64_t windowBegin;  // output positions [windowBegin, windowEnd) of this batch
  uint64_t windowEnd;
  int32_t joinType;
  int32_t* mapping;
  int32_t* buildRows;
  const uint2* staged;       // sparse listing: {probe row, build row} per tile, kSparseCap apart
  const uint8_t* tileDense;  // tiles that fell back to hits[]
  JoinFilterArgs f;          // extra filter: only the passing rows of a chain are listed
  // build_rows_out of a probe row without a match / with a null key: -1, or -2 where the
  // null-aware semi project join says NULL (HashProbe::fillLeftSemiProjectMatchColumn)
  int32_t missValue;
  int32_t nullKeyValue;
  // inline dependents (inner joins over wide slots): output column i = the value staged next to
  // the hit; listed tiles take it from the dependent column by build row
  const uint64_t* hitVals[kWideDeps];
  const uint64_t* depVals[kWideDeps];
  uint64_t* outVals[kWideDeps];
  // output rows of every quarter tile, when the probe pass counted them (k_join_probe_grouped): the
  // 0-or-1-rows-per-probe-row path takes its wave totals from here instead of reading hits[] twice
  const uint32_t* quarterSums;
};
static_assert(sizeof(EmitArgs) <= 4096, "kernel arguments are limited to 4 KB");

// Each block owns one tile of probe rows and walks it in 256-row steps: a block
// scan of the per-row output counts gives every row its output offset; rows
// intersecting the window write their pairs. Probe rows ascend with the
// offsets, so the output is in ascending probe-row order with all matches of a
// row contiguous.
__global__ __launch_bounds__(256) void k_emit(EmitArgs a) {
  __shared__ uint64_t waveTotals[4];
  __shared__ uint64_t running;
  const int64_t tile = a.firstTile + blockIdx.x;
  if (a.staged && !a.tileDense[tile]) {
    // The probe pass already listed this tile's hits in probe-row order.
    const uint64_t off = a.tileOffsets[tile];
    const uint64_t k = a.tileOffsets[tile + 1] - off;
    const bool wantBuild = listsMatches(a.joinType);
    for (uint64_t i = threadIdx.x; i < k; i += blockDim.x) {
      const uint64_t pos = off + i;
      if (pos >= a.windowBegin && pos < a.windowEnd) {
        const uint2 pair = a.staged[tile * kSparseCap + i];
        a.mapping[pos - a.windowBegin] = static_cast<int32_t>(pair.x);
        if (a.buildRows) {
          a.buildRows[pos - a.windowBegin] = wantBuild ? static_cast<int32_t>(pair.y) : -1;
        }
#pragma unroll
        for (int d = 0; d < kWideDeps; ++d) {
          if (a.outVals[d]) {
            a.outVals[d][pos - a.windowBegin] = a.depVals[d][pair.y];
          }
        }
      }
    }
    return;
  }
  if (a.counts == nullptr) {
    // Every probe row produces 0 or 1 output rows: ballot compaction. Each wave
    // owns 2048 consecutive rows of the tile; one pass counts, one block-level
    // exchange of the four wave totals, one pass writes.
    constexpr int kWaveRows = kTileRows / 4;
    const int64_t waveBase = tile * kTileRows + static_cast<int64_t>(threadIdx.x >> 6) * kWaveRows;
    uint64_t total = 0;
    if (a.quarterSums) {
      total = a.quarterSums[tile * 4 + (threadIdx.x >> 6)];
    } else {
      for (int it = 0; it < kWaveRows / 64; ++it) {
        const int64_t r = waveBase + it * 64 + lane();
        const bool out = r < a.numRows &&
            outputCount(a.joinType, isHit(a.hits[r]) ? 1 : 0, a.hits[r] == kNullKey32) != 0;
        total += popc64(ballot(out));
      }
    }
    if (lane() == 0) {
      waveTotals[threadIdx.x >> 6] = total;
    }
    blockSync();
    uint64_t run = a.tileOffsets[tile];
    for (int w = 0; w < (threadIdx.x >> 6); ++w) {
      run += waveTotals[w];
    }
    if (total == 0 || run >= a.windowEnd || run + total <= a.windowBegin) {
      return;
    }
    for (int it = 0; it < kWaveRows / 64; ++it) {
      const int64_t r = waveBase + it * 64 + lane();
      uint32_t hit = kNoRow32;
      bool out = false;
      if (r < a.numRows) {
        hit = a.hits[r];
        out = outputCount(a.joinType, isHit(hit) ? 1 : 0, hit == kNullKey32) != 0;
      }
      const uint64_t m = ballot(out);
      if (m == 0) {
        continue;
      }
      const uint64_t pos = run + lanePrefix(m);
      if (out && pos >= a.windowBegin && pos < a.windowEnd) {
        a.mapping[pos - a.windowBegin] = static_cast<int32_t>(r);
        if (a.buildRows) {
          const bool listMatch = isHit(hit) &&
              (listsMatches(a.joinType) || a.joinType == VX355_JOIN_LEFT_SEMI_PROJECT);
          a.buildRows[pos - a.windowBegin] =
              listMatch ? static_cast<int32_t>(hit) : (hit == kNullKey32 ? a.nullKeyValue : a.missValue);
        }
#pragma unroll
        for (int d = 0; d < kWideDeps; ++d) {
          if (a.outVals[d]) {
            a.outVals[d][pos - a.windowBegin] = a.hitVals[d][r];
          }
        }
      }
      run += popc64(m);
    }
    return;
  }
  if (threadIdx.x == 0) {
    running = a.tileOffsets[tile];
  }
  blockSync();
  for (int step = 0; step < kTileRows / 256; ++step) {
    const int64_t r = tile * kTileRows + step * 256 + threadIdx.x;
    uint32_t hit = kNoRow32;
    uint32_t c = 0;
    if (r < a.numRows) {
      hit = a.hits[r];
      c = a.counts[r];
    }
    uint64_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint64_t v = shfl64(incl, lane() - off >= 0 ? lane() - off : lane());
      if (lane() >= off) {
        incl += v;
      }
    }
    if (lane() == 63) {
      waveTotals[threadIdx.x >> 6] = incl;
    }
    blockSync();
    uint64_t lo = running + (incl - c);
    for (int w = 0; w < (threadIdx.x >> 6); ++w) {
      lo += waveTotals[w];
    }
    const uint64_t stepTotal = waveTotals[0] + waveTotals[1] + waveTotals[2] + waveTotals[3];
    const uint64_t hi = lo + c;
    if (c != 0 && hi > a.windowBegin && lo < a.windowEnd) {
      const bool listMatches = isHit(hit) && listsMatches(a.joinType);
      const bool firstOnly = isHit(hit) && a.joinType == VX355_JOIN_LEFT_SEMI_PROJECT;
      uint32_t b = hit;
      for (uint64_t p = lo; p < hi && p < a.windowEnd; ++p) {
        if (listMatches && a.f.numTerms) {
          // hits[] holds the first passing row (k_join_filter); later ones are found by re-evaluating
          while (p != lo && !evalJoinFilter(a.f, r, b)) {
            b = a.next[b];
          }
        }
        if (p >= a.windowBegin) {
          a.mapping[p - a.windowBegin] = static_cast<int32_t>(r);
          if (a.buildRows) {
            a.buildRows[p - a.windowBegin] = (listMatches || firstOnly)
                ? static_cast<int32_t>(b)
                : (hit == kNullKey32 ? a.nullKeyValue : a.missValue);
          }
        }
        if (listMatches) {
          b = a.next[b];
        }
      }
    }
    blockSync();
    if (threadIdx.x == 0) {
      running += stepTotal;
    }
    blockSync();
    if (running >= a.windowEnd) {
      break;  // uniform: 'running' is shared
    }
  }
}

// Validity words of n rows that are all valid (what k_gather_deps writes for a column without nulls).
__global__ __launch_bounds__(256) void k_fill_valid(uint64_t* words, int64_t n) {
  const int64_t w = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (w * 64 < n) {
    const int64_t rest = n - w * 64;
    words[w] = rest >= 64 ? ~0ULL : ((1ULL << rest) - 1);
  }
}

struct GatherArgs {
  const int32_t* buildRows;
  int32_t count;
  int32_t numCols;
  const char* depVals[kMaxDeps];
  const uint8_t* depValid[kMaxDeps];
  int32_t width[kMaxDeps];
  int32_t kind[kMaxDeps];
  void* outVals[kMaxDeps];
  uint64_t* outNulls[kMaxDeps];
};

// extractColumns (HashProbe.cpp:82-118): build columns at the listed rows.
__global__ __launch_bounds__(256) void k_gather_deps(GatherArgs a) {
  const int32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos - static_cast<int32_t>(lane()) >= a.count) {
    return;  // whole wave past the end: its bitmap word lies outside ceil(count / 64) words
  }
  const bool active = pos < a.count;
  const int32_t b = active ? a.buildRows[pos] : -1;
  for (int c = 0; c < a.numCols; ++c) {
    const bool valid = active && b >= 0 && (a.depValid[c] == nullptr || a.depValid[c][b] != 0);
    uint64_t m = ballot(valid);
    if (a.outNulls[c] && lane() == 0) {
      a.outNulls[c][pos >> 6] = m;
    }
    const int w = a.width[c];
    if (w == 0) {
      uint64_t bits = ballot(valid && a.depVals[c][b] != 0);
      if (lane() == 0) {
        static_cast<uint64_t*>(a.outVals[c])[pos >> 6] = bits;
      }
      continue;
    }
    if (!active) {
      continue;
    }
    char* dst = static_cast<char*>(a.outVals[c]) + static_cast<int64_t>(pos) * w;
    if (!valid) {
      for (int i = 0; i < w; ++i) {
        dst[i] = 0;
      }
      continue;
    }
    const char* src = a.depVals[c] + static_cast<int64_t>(b) * w;
    if (w == 8) {
      *reinterpret_cast<uint64_t*>(dst) = *reinterpret_cast<const uint64_t*>(src);
    } else if (w == 4) {
      *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(src);
    } else if (w == 16) {
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
    } else {
      for (int i = 0; i < w; ++i) {
        dst[i] = src[i];
      }
    }
  }
}

bool supportedJoin(int32_t t) { return t >= VX355_JOIN_INNER && t <= VX355_JOIN_RIGHT_ANTI; }

// HashBuild.cpp:257-268: these kinds list build rows, so rows with null keys stay in the table.
bool keepsNullKeyRows(int32_t t) {
  return t == VX355_JOIN_RIGHT || t == VX355_JOIN_FULL || t == VX355_JOIN_RIGHT_SEMI_PROJECT ||
      t == VX355_JOIN_RIGHT_ANTI;
}
// Rows with null keys stay in the row store (and, with nullAsValue, enter the table).
// ... and in the row store of a null-aware join: with an extra filter its null-key build rows
// take part in the result (HashProbe::evalFilterForNullAwareJoin, HashProbe.cpp:1639-1700).
bool retainsNullKeyRows(int32_t joinType, bool nullAsValue, bool nullAware = false) {
  return keepsNullKeyRows(joinType) || nullAsValue || nullAware;
}
bool marksProbedRows(int32_t t) {
  return t == VX355_JOIN_RIGHT || t == VX355_JOIN_FULL || t == VX355_JOIN_RIGHT_SEMI_FILTER ||
      t == VX355_JOIN_RIGHT_SEMI_PROJECT || t == VX355_JOIN_RIGHT_ANTI;
}
bool countingJoin(int32_t t) {
  return t == VX355_JOIN_COUNTING_LEFT_SEMI_FILTER || t == VX355_JOIN_COUNTING_ANTI;
}

// bit r = (probed[r] != 0) == wantProbed
__global__ __launch_bounds__(256) void k_probed_bits(const uint8_t* probed, int64_t n, int32_t wantProbed,
                                                      uint64_t* bits) {
  const int64_t numWords = (n + 63) >> 6;
  const int64_t waveStride = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t w = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; w < numWords;
       w += waveStride) {
    const int64_t r = (w << 6) + lane();
    const bool on = r < n && (wantProbed == 2 || (probed[r] != 0) == (wantProbed != 0));  // 2 = every row
    const uint64_t word = ballot(on);
    if (lane() == 0) {
      bits[w] = word;
    }
  }
}

}  // namespace
}  // namespace vx

using namespace vx;

struct vx355_join_build {
  vx::Runtime* ctx = nullptr;  // this operator's execution context (stream, mailbox)
  vx::AsyncQueue* aq = nullptr;  // worker of vx355_join_build_add_input_async (created on first use)
  std::vector<int32_t> keyCols, keyKinds, depCols, depKinds;
  std::vector<int32_t> usedCols;
  int32_t joinType = 0;
  std::vector<DevBuf> keyVals;   // key images per row: 1 word, 2 for strings / timestamps
  bool unmappable = false;       // some key has no 64-bit normalized form
  std::vector<DevBuf> depVals;   // width bytes per row (BOOLEAN: 1 byte)
  std::vector<DevBuf> depValid;  // 1 byte per row
  uint32_t depNulls = 0;         // bit d: column d holds a null
  int64_t numRows = 0;
  int64_t capacityRows = 0;
  bool hasNullKeys = false;
  bool finished = false;
  DevBuf keyNull;  // right / full joins: 1 byte per row, set for rows with a null key (nullAsValue: mask of null keys)
  bool nullAsValue = false;  // HashJoinNode::isNullAsValue
  bool dropDuplicates = false;  // HashJoinNode::canDropDuplicates
  bool nullAware = false;    // HashJoinNode::isNullAware
  std::vector<int64_t> obsMin, obsMax;
  DevBuf countersBuf, scratch, validWords, rowList;
  // arena of key / payload strings longer than 12 bytes (blocks never move: stored views point into them)
  std::vector<DevBuf> strBlocks;
  DevBuf strCursor;
  uint64_t strCap = 0, strUsed = 0;
  bool hasStringCols = false;
  HostCoalescer coalescer;  // small host batches -> large appends
};

struct vx355_join_table {
  std::atomic<int> refs{1};
  int device = -1;        // the GPU the table lives on
  std::mutex lazyMutex;   // dynamic-filter state computed on first request
  int32_t mode = JMODE_ARRAY;
  int32_t joinType = 0;
  bool droppedDuplicates = false;  // built with drop_duplicates: one linked row per key
  std::vector<int32_t> keyKinds, depKinds;
  std::vector<KeyRange> ranges;
  uint64_t capacity = 0;
  DevBuf head;   // array mode: u32[capacity], only entries whose presence bit is set are valid
  DevBuf present;  // array mode: bit per possible key
  DevBuf slots;  // hash mode: Slot[capacity], or WideSlot[capacity] when slotShift == 5
  int32_t slotShift = 4;
  uint64_t homeMask = 0, wrapMask = 0;  // see nextSlot (join.hip)
  DevBuf next;   // u32[numRows]
  DevBuf gslots;  // generic hash mode: u64[capacity]
  std::vector<DevBuf> keyStore;  // generic hash mode: build key images
  std::vector<DevBuf> depVals, depValid;
  uint32_t depNulls = 0;   // bit d: dependent column d holds nulls (else depValid[d] is all ones and nobody reads it)
  // slots with inline dependents (WideSlot), built by the first probe batch large enough to pay
  // for it (ensureWide, under lazyMutex); wideDeps = the dependents they carry
  DevBuf wide;
  int32_t wideDeps[kWideDeps] = {-1, -1};
  int32_t numWide = 0;
  bool wideTried = false;
  int64_t numRows = 0;
  int64_t numDistinct = 0;
  bool hasDuplicates = false;
  bool hasNullKeys = false;
  DevBuf probed;   // right / full / right semi / right anti joins: 1 byte per build row
  DevBuf remaining;  // counting joins: occurrences left per distinct key, at the chain's head row
  std::vector<DevBuf> strBlocks;  // the builds' string arenas
  bool keepsNullRows = false;
  bool nullAsValue = false;
  DevBuf keyNull;  // nullAsValue: the builds' null-key masks, by build row
  DevBuf nullKeyRows;         // null-aware joins: the build rows whose key is null (u32 list)
  int64_t numNullKeyRows = 0;
  // dynamic filters: ascending distinct values per key, computed on first request
  std::vector<DevBuf> distinctVals;
  std::vector<int64_t> distinctCount;  // -1 = not computed
};

struct vx355_join_probe {
  vx::Runtime* ctx = nullptr;  // this operator's execution context (stream, mailbox)
  vx::AsyncQueue* aq = nullptr;  // worker of vx355_join_probe_add_input_async (created on first use)
  // pages of vx355_join_probe_get_output_async by ticket, until vx355_join_probe_output_result hands them out
  struct QueuedPage {
    std::vector<vx355_out_column> cols;
    std::vector<int32_t> ids;
    int32_t numRows = 0, finished = 0;
    int status = VX355_OK;
    std::string errorText;
    std::atomic<bool> complete{false};
  };
  std::mutex pagesMutex;
  std::map<int64_t, std::shared_ptr<QueuedPage>> pages;
  vx355_join_table* table = nullptr;
  std::vector<int32_t> keyCols;
  int32_t joinType = 0;
  DevBuf hits, counts, tileSums, tileOffsets, scratch, outMap, outRows;
  DevBuf staged, tileDense, sparseStats;  // sparse listing (probeAddInput)
  // range-partitioned probe: records, histogram, offsets, hit words (unsorted / sorted)
  DevBuf ppRecs, ppHist, ppOffsets, ppScan, ppPairs, ppSorted, ppSortTmp;
  bool pairList = false;     // the batch's output is ppSorted[0, totalOut)
  int64_t outputBatchBytes = 0;  // preferred_output_batch_bytes (0 = rows only)
  int32_t partitionMode = -1;  // VX355_JOIN_PARTITION: -1 adaptive, 0 never, 1 whenever eligible
  bool window = true;          // VX355_JOIN_WINDOW=0: the listing probe gathers every presence word itself
  int32_t wideMode = -1;       // VX355_JOIN_WIDE: -1 adaptive (large batches), 0 never, 1 whenever eligible
  DevBuf hitVals[kWideDeps];   // inline dependents of the batch's hits, by probe row
  int32_t wideStaged = 0;      // how many of the table's wide dependents this batch staged
  int32_t denseStreak = 0;     // batches that still skip the sparse-listing sample
  bool partitionFast = true;   // VX355_JOIN_PARTITION_FAST=0: the range-partitioned probe always counts first
  // vx355_join_probe_add_input_regrouped: VX355_JOIN_REGROUP -1 adaptive, 0 never, 1 whenever eligible;
  // VX355_JOIN_SLICE_BYTES = bytes of the slot array one group of rows is probed against;
  // VX355_JOIN_GROUP_WGS = workgroups per CU of the grouped probe (0: from the rows per slice)
  int32_t regroupMode = -1;
  int64_t sliceBytes = 4 << 20;   // (an XCD's L2; 2 MiB: probe 2 % faster, regrouping 10-15 % slower - twice the bins)
  int32_t groupWgs = 0;
  bool groupPrefetch = false;  // VX355_JOIN_GROUP_PREFETCH=1: the grouped probe streams its share of the slice first (measured: no gain)
  int32_t groupUnit = 2048;    // VX355_JOIN_GROUP_UNIT: rows per workgroup of the grouped probe (2048 or 8192)
  DevBuf grpHist, grpOffsets, grpScan, grpUnits, grpXcd, grpUnitSums;
  bool haveUnitSums = false;   // the batch's output rows per quarter tile are in grpUnitSums
  DeviceBatch batch;                       // the batch being probed: the filter reads it at output time
  std::vector<std::vector<char>> hostStrings;  // long payload strings of the last page handed to a host caller
  std::vector<vx355_join_filter_term> filter;
  std::vector<vx355_filter_term> inputFilter;  // fused FilterProject in front of the probe (set_input_filter)
  std::vector<int32_t> usedCols;           // key columns + the filters' probe columns
  DevBuf hitBits, hitRows, hitWords, hitSorted, sortTmp;  // counting joins
  bool sparse = false;
  int32_t sparseMode = -1;   // VX355_JOIN_SPARSE: -1 adaptive, 0 never, 1 always
  int64_t sparseTiles = 0;   // statistics of the last batch: tiles listed by the probe pass
  int64_t denseTiles = 0;
  std::vector<uint64_t> hostTileOffsets;
  int64_t numRows = 0;
  int64_t numTiles = 0;
  uint64_t totalOut = 0;
  uint64_t cursor = 0;
  bool hasInput = false;
  bool haveCounts = false;
  bool nullAware = false;
  int32_t missValue = -1, nullKeyValue = -1;  // see EmitArgs
  // build-side output (right / full / right semi): the listed rows, computed on first request
  DevBuf buildSideRows;
  int64_t buildSideCount = -1;
  int64_t buildSideCursor = 0;
};

namespace vx {
namespace {

int keyWordsOf(int32_t kind) { return (isString(kind) || kind == VX355_TIMESTAMP) ? 2 : 1; }

int depStoreWidth(int32_t kind) {
  int w = kindWidth(kind);
  return w == 0 ? 1 : w;
}

void growBuild(vx355_join_build& h, int64_t rows) {
  if (rows <= h.capacityRows) {
    return;
  }
  int64_t cap = std::max<int64_t>(rows, h.capacityRows * 2);
  cap = std::max<int64_t>(cap, 1024);
  for (size_t k = 0; k < h.keyVals.size(); ++k) {
    const size_t w = static_cast<size_t>(keyWordsOf(h.keyKinds[k])) * 8;
    h.keyVals[k].ensure(static_cast<size_t>(cap) * w + 64, true, static_cast<size_t>(h.numRows) * w);
  }
  if (retainsNullKeyRows(h.joinType, h.nullAsValue, h.nullAware)) {
    h.keyNull.ensure(static_cast<size_t>(cap) + 64, true, static_cast<size_t>(h.numRows));
  }
  for (size_t d = 0; d < h.depVals.size(); ++d) {
    const int w = depStoreWidth(h.depKinds[d]);
    h.depVals[d].ensure(static_cast<size_t>(cap) * w + 64, true, static_cast<size_t>(h.numRows) * w);
    h.depValid[d].ensure(static_cast<size_t>(cap) + 64, true, static_cast<size_t>(h.numRows));
  }
  h.capacityRows = cap;
}

void resetBuildCounters(DevBuf& buf) {
  BuildCounters c{};
  for (int k = 0; k < kMaxKeys; ++k) {
    c.keyMin[k] = INT64_MAX;
    c.keyMax[k] = INT64_MIN;
  }
  copyIn(buf.ensure(sizeof(BuildCounters)), &c, VX355_MEM_HOST, sizeof(c));
}

BuildCounters readBuildCounters(DevBuf& buf) {
  BuildCounters c;
  copyOut(&c, VX355_MEM_HOST, buf.ptr(), sizeof(c));
  return c;
}

void buildAddInput(vx355_join_build& h, const vx355_batch* batch) {
  VX_CHECK_ARG(!h.finished, "addInput after finish");
  DeviceBatch db;
  db.load(batch, h.usedCols);
  const int64_t n = db.numRows();
  if (n == 0) {
    return;
  }
  resetBuildCounters(h.countersBuf);
  auto* ctr = h.countersBuf.as<BuildCounters>();
  const int64_t words = ceilDiv(n, 64);
  ValidArgs va{};
  va.numKeys = static_cast<int32_t>(h.keyCols.size());
  for (int k = 0; k < va.numKeys; ++k) {
    va.keys[k] = db.col(h.keyCols[k]);
  }
  va.numRows = n;
  bool anyKeyNulls = false;
  for (int k = 0; k < va.numKeys; ++k) {
    anyKeyNulls = anyKeyNulls || va.keys[k].nulls != nullptr;
  }
  const bool keepNulls = retainsNullKeyRows(h.joinType, h.nullAsValue, h.nullAware);
  int32_t* rows = nullptr;
  int64_t selected = n;
  if (anyKeyNulls) {  // key columns without null bitmaps (the common case) skip both passes
    va.validWords = static_cast<uint64_t*>(h.validWords.ensure(static_cast<size_t>(words) * 8 + 64));
    va.counters = ctr;
    VX_LAUNCH("k_key_valid", k_key_valid, streamGrid(words * 64, 256), 256, 0, va);
    if (!keepNulls) {
      rows = static_cast<int32_t*>(h.rowList.ensure(static_cast<size_t>(n) * 4 + 64));
      compactBits(va.validWords, nullptr, nullptr, n, rows, h.scratch, &selected);
    }
  }
  if (selected > 0) {
    growBuild(h, h.numRows + selected);
    AppendArgs aa{};
    aa.numKeys = va.numKeys;
    aa.numDeps = static_cast<int32_t>(h.depCols.size());
    for (int k = 0; k < aa.numKeys; ++k) {
      aa.keys[k] = va.keys[k];
      aa.keyOut[k] = h.keyVals[k].as<uint64_t>();
      aa.keyWords[k] = keyWordsOf(h.keyKinds[k]);
    }
    for (int d = 0; d < aa.numDeps; ++d) {
      aa.deps[d] = db.col(h.depCols[d]);
      aa.depOut[d] = h.depVals[d].as<char>();
      aa.depValid[d] = h.depValid[d].as<uint8_t>();
      aa.depWidth[d] = kindWidth(h.depKinds[d]);
    }
    aa.rows = rows;
    aa.count = selected;
    aa.base = h.numRows;
    aa.counters = ctr;
    if (h.hasStringCols) {
      auto& rt = Runtime::get();
      unsigned long long* cursor = static_cast<unsigned long long*>(h.strCursor.ensure(64));
      HIP_OK(hipMemsetAsync(cursor + 1, 0, 8, rt.stream));
      VX_LAUNCH("k_build_long_bytes", k_build_long_bytes, streamGrid(selected, 256), 256, 0, aa, cursor + 1);
      unsigned long long need = 0;
      copyOut(&need, VX355_MEM_HOST, cursor + 1, 8);
      if (need > 0) {
        if (h.strBlocks.empty() || h.strUsed + need > h.strCap) {
          h.strBlocks.emplace_back();
          h.strCap = std::max<uint64_t>(need, 16ULL << 20);
          h.strBlocks.back().ensure(static_cast<size_t>(h.strCap) + 64);
          h.strUsed = 0;
          HIP_OK(hipMemsetAsync(cursor, 0, 8, rt.stream));
        }
        h.strUsed += need;
        aa.arenaBase = h.strBlocks.back().as<char>();
        aa.arenaCursor = cursor;
        aa.arenaCap = h.strCap;
      }
    }
    if (!aa.arenaCursor) {
      aa.arenaCursor = static_cast<unsigned long long*>(h.strCursor.ensure(64));  // never dereferenced usefully
    }
    if (keepNulls) {
      aa.keyValidWords = anyKeyNulls ? va.validWords : nullptr;
      aa.keyNullOut = h.keyNull.as<uint8_t>();
    }
    aa.nullAsValue = h.nullAsValue ? 1 : 0;
    if (!appendFlat(aa)) {
      VX_LAUNCH("k_build_append", k_build_append, streamGrid(selected, 256), 256, 0, aa);
    }
  }
  BuildCounters c = readBuildCounters(h.countersBuf);
  if (c.unmappable) {
    h.unmappable = true;  // finish() picks the generic hash mode
  }
  if (c.longString) {
    VX_THROW(VX355_EINTERNAL, "build-side string arena exhausted");
  }
  if (c.nullKeyRows) {
    h.hasNullKeys = true;
  }
  h.depNulls |= c.depNulls;
  for (size_t k = 0; k < h.keyCols.size(); ++k) {
    if (c.keyMin[k] <= c.keyMax[k]) {
      h.obsMin[k] = std::min(h.obsMin[k], c.keyMin[k]);
      h.obsMax[k] = std::max(h.obsMax[k], c.keyMax[k]);
    }
  }
  h.numRows += selected;
}

void launchGroupScatter(const GroupArgs& g, int tiles) {
  auto& rt = Runtime::get();
  // The sorted sub-tile goes through a stage of half its rows at a time (64 KB of LDS in all up to 1024 bins:
  // two workgroups per CU, one's loads behind the other's LDS phases): 2.75 instead of 3.15 ms per 200 M
  // rows of two 8-byte columns in the same run (profiles/r06_c5_variants.md). VX355_JOIN_SCATTER_HALF=0: the
  // whole sub-tile at once, one workgroup per CU.
  static const bool half = !(std::getenv("VX355_JOIN_SCATTER_HALF") && std::atoi(std::getenv("VX355_JOIN_SCATTER_HALF")) == 0);
  const size_t lds = half ? groupScatterLds<true>(g.numBins) : groupScatterLds<false>(g.numBins);
  auto kernel = half ? k_grp_scatter<true> : k_grp_scatter<false>;
  if (lds > (48u << 10)) {
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(lds)));
  }
  const int perCu = std::max<int>(1, static_cast<int>((160u << 10) / (lds + 1024)));
  VX_LAUNCH("k_grp_scatter", kernel, std::min<int>(tiles, rt.numCUs * std::min(perCu, 2)), 1024, lds, g);
}

// The normalized-key table of 't' (capacity, slotShift, homeMask set; slots allocated) assembled
// from the rows of 'h' group by group in LDS (see k_lds_build). false: some group overflowed (more
// distinct keys than slots hash into 128 KB of the table - never with a mixing hash and load <= 0.7,
// but not impossible): the caller inserts the classic way.
bool buildInLds(vx355_join_build& h, vx355_join_table& t, const InsertArgs& ia) {
  auto& rt = Runtime::get();
  const int64_t n = h.numRows;
  const uint64_t cap = t.capacity;
  const int slotShift = t.slotShift;
  // 64 KB of slots per group: two workgroups share a CU, one streams records while the other
  // writes its image out
  uint64_t groupBytes = 64 << 10;
  if (const char* e = std::getenv("VX355_JOIN_LDS_GROUP_BYTES")) {
    groupBytes = nextPow2(std::min<uint64_t>(std::max<uint64_t>(std::atoll(e), 4096), 128 << 10));
  }
  const uint64_t groupSlots = std::min<uint64_t>(cap, groupBytes >> slotShift);
  const uint64_t numGroups = cap / groupSlots;
  const uint64_t bins = std::min<uint64_t>(numGroups, kGrpMaxBins);
  DevBuf normKeys, recs, hist, offsetsBuf, scan;
  uint64_t* keys = static_cast<uint64_t*>(normKeys.ensure(static_cast<size_t>(n) * 8 + 64));
  VX_LAUNCH("k_norm_keys", k_norm_keys, streamGrid(n, 256), 256, 0, ia, keys);
  GroupArgs g{};
  g.numKeys = 1;
  g.keys[0].values = keys;
  g.keys[0].kind = VX355_BIGINT;
  g.keys[0].enc = VX355_FLAT;
  g.ranges[0].min = 1;   // the column already holds normalized keys: id = v - 1 + 1
  g.ranges[0].max = INT64_MAX;
  g.ranges[0].multiplier = 1;
  g.fastKey = 1;
  g.numBins = static_cast<int32_t>(bins);
  g.mask = cap - 1;
  g.binShift = __builtin_ctzll(cap) - __builtin_ctzll(bins);
  g.numRows = n;
  g.numTiles = ceilDiv(n, kGrpTileRows);
  char* records = static_cast<char*>(recs.ensure((static_cast<size_t>(n) << slotShift) + 64));
  auto part = [&](const void* in, int stride, int offset, int width) {
    const int q = g.numParts++;
    g.in[q] = static_cast<const char*>(in);
    g.stride[q] = stride;
    g.out[q] = records + offset;
    g.outStride[q] = 1 << slotShift;
    g.width[q] = width;
  };
  part(keys, 8, 0, 8);        // part 0 = the key column (k_grp_scatter takes the bins from it)
  part(nullptr, 0, 8, 4);     // the row number
  if (slotShift == 5) {
    part(ia.wideDep[0], 8, 16, 8);
    if (ia.wideDep[1]) {
      part(ia.wideDep[1], 8, 24, 8);
    }
  }
  g.numCols = g.numParts;
  const int64_t cells = g.numTiles * g.numBins;
  g.hist = static_cast<uint32_t*>(hist.ensure(static_cast<size_t>(cells) * 4 + 64));
  uint64_t* offsets = static_cast<uint64_t*>(offsetsBuf.ensure(static_cast<size_t>(cells + 1) * 8 + 64));
  g.offsets = offsets;
  const int grid = static_cast<int>(std::min<int64_t>(g.numTiles, static_cast<int64_t>(rt.numCUs) * 2));
  VX_LAUNCH("k_grp_hist", k_grp_hist, grid, 1024, 0, g);
  scanU32ToU64(g.hist, cells, offsets, scan);
  launchGroupScatter(g, grid);
  LdsBuildArgs b{};
  b.recs = records;
  b.offsets = offsets;
  b.histTiles = g.numTiles;
  b.groupsPerBinShift = __builtin_ctzll(numGroups) - __builtin_ctzll(bins);
  b.slotShift = slotShift;
  b.groupShift = __builtin_ctzll(groupSlots);
  b.dropDups = ia.dropDups;
  b.mask = cap - 1;
  b.homeMask = t.homeMask;
  b.slots = t.slots.as<char>();
  b.next = ia.next;
  b.counters = ia.counters;
  b.numGroups = static_cast<int32_t>(numGroups);
  const size_t lds = static_cast<size_t>(groupSlots) << slotShift;
  if (lds > (48u << 10)) {
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds_build), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(lds)));
  }
  const int blocks = static_cast<int>(ceilDiv(static_cast<int64_t>(numGroups), 8) * 8);
  VX_LAUNCH("k_lds_build", k_lds_build, blocks, 1024, lds, b);
  BuildCounters c = readBuildCounters(h.countersBuf);
  if (c.tableFull) {
    resetBuildCounters(h.countersBuf);
    return false;
  }
  t.wrapMask = groupSlots - 1;
  return true;
}

vx355_join_table* buildFinish(vx355_join_build& h, vx355_join_build* const* others, int32_t numOthers) {
  auto& rt = Runtime::get();
  VX_CHECK_ARG(!h.finished, "finish called twice");
  // Merge the peer Drivers' rows behind ours (HashBuild.cpp:903-922): row ids
  // are [this build's rows..., others[0]'s rows..., ...].
  int64_t total = h.numRows;
  for (int32_t i = 0; i < numOthers; ++i) {
    VX_CHECK_ARG(others && others[i] && others[i] != &h, "bad peer build handle");
    VX_CHECK_ARG(others[i]->keyKinds == h.keyKinds && others[i]->depKinds == h.depKinds &&
                     others[i]->joinType == h.joinType,
                 "peer build with a different layout");
    total += others[i]->numRows;
  }
  if (total > static_cast<int64_t>(INT32_MAX)) {
    // build row ids travel as int32 (build_rows_out) and as u32 with two sentinels (next[], head[])
    VX_THROW(VX355_EUNSUPPORTED, "more than 2^31-1 build rows in one table");
  }
  growBuild(h, total);
  for (int32_t i = 0; i < numOthers; ++i) {
    auto& o = *others[i];
    if (o.numRows > 0) {
      for (size_t k = 0; k < h.keyVals.size(); ++k) {
        const size_t w = static_cast<size_t>(keyWordsOf(h.keyKinds[k])) * 8;
        copyIn(h.keyVals[k].as<char>() + h.numRows * w, o.keyVals[k].ptr(), VX355_MEM_DEVICE,
               static_cast<size_t>(o.numRows) * w);
      }
      if (retainsNullKeyRows(h.joinType, h.nullAsValue, h.nullAware)) {
        copyIn(h.keyNull.as<char>() + h.numRows, o.keyNull.ptr(), VX355_MEM_DEVICE, static_cast<size_t>(o.numRows));
      }
      h.unmappable = h.unmappable || o.unmappable;
      for (size_t d = 0; d < h.depVals.size(); ++d) {
        const int w = depStoreWidth(h.depKinds[d]);
        copyIn(h.depVals[d].as<char>() + h.numRows * w, o.depVals[d].ptr(), VX355_MEM_DEVICE,
               static_cast<size_t>(o.numRows) * w);
        copyIn(h.depValid[d].as<char>() + h.numRows, o.depValid[d].ptr(), VX355_MEM_DEVICE,
               static_cast<size_t>(o.numRows));
      }
      for (size_t k = 0; k < h.keyCols.size(); ++k) {
        h.obsMin[k] = std::min(h.obsMin[k], o.obsMin[k]);
        h.obsMax[k] = std::max(h.obsMax[k], o.obsMax[k]);
      }
      h.numRows += o.numRows;
    }
    h.hasNullKeys = h.hasNullKeys || o.hasNullKeys;
    h.depNulls |= o.depNulls;
    o.finished = true;
  }
  rt.sync();

  auto t = std::make_unique<vx355_join_table>();
  t->device = rt.device;
  t->joinType = h.joinType;
  t->keyKinds = h.keyKinds;
  t->depKinds = h.depKinds;
  t->numRows = h.numRows;
  t->hasNullKeys = h.hasNullKeys;
  t->ranges.resize(h.keyCols.size());

  // decideHashMode for a join build: exact ranges (reserve 0).
  unsigned __int128 product = 1;
  bool overflow = false;
  for (size_t k = 0; k < h.keyCols.size(); ++k) {
    KeyRange& r = t->ranges[k];
    if (h.keyKinds[k] == VX355_BOOLEAN) {
      r.min = 0;
      r.max = 1;
      r.rangeSize = 3;
    } else if (h.obsMin[k] > h.obsMax[k]) {
      r.min = 0;
      r.max = -1;
      r.rangeSize = 1;
    } else {
      int64_t span;
      if (__builtin_sub_overflow(h.obsMax[k], h.obsMin[k], &span) || span >= (1LL << 59) - 1) {
        overflow = true;
        span = 0;
      }
      r.min = h.obsMin[k];
      r.max = h.obsMax[k];
      r.rangeSize = static_cast<uint64_t>(span) + 2;
    }
    r.multiplier = static_cast<uint64_t>(product);
    product *= r.rangeSize;
    if (product >= (static_cast<unsigned __int128>(1) << 64) - 1) {
      overflow = true;
    }
  }
  const bool generic = overflow || h.unmappable;  // decideHashMode's kHash cases
  // Direct addressing while the head array stays within a budget relative to
  // the build size (4 bytes per possible key).
  uint64_t arrayMax = std::max<uint64_t>(1ULL << 21, static_cast<uint64_t>(h.numRows) * 64);
  if (const char* e = std::getenv("VX355_JOIN_ARRAY_MAX")) {
    arrayMax = std::strtoull(e, nullptr, 10);
  }
  const uint64_t range = generic ? 0 : static_cast<uint64_t>(product);
  InsertArgs ia{};
  ia.numKeys = static_cast<int32_t>(h.keyCols.size());
  for (int k = 0; k < ia.numKeys; ++k) {
    ia.keyStore[k] = h.keyVals[k].as<uint64_t>();
    ia.keyWords[k] = keyWordsOf(h.keyKinds[k]);
    ia.keyIsString[k] = isString(h.keyKinds[k]) ? 1 : 0;
    ia.keyKind[k] = h.keyKinds[k];
    ia.ranges[k] = t->ranges[k];
  }
  ia.numRows = h.numRows;
  if (retainsNullKeyRows(h.joinType, h.nullAsValue, h.nullAware) && h.hasNullKeys) {
    ia.keyNull = h.keyNull.as<uint8_t>();
    t->keepsNullRows = true;
  }
  ia.nullAsValue = h.nullAsValue ? 1 : 0;
  ia.dropDups = h.dropDuplicates ? 1 : 0;
  t->nullAsValue = h.nullAsValue;
  t->droppedDuplicates = h.dropDuplicates;
  resetBuildCounters(h.countersBuf);
  ia.counters = h.countersBuf.as<BuildCounters>();
  t->next.ensure(static_cast<size_t>(std::max<int64_t>(1, h.numRows)) * 4 + 64);
  ia.next = t->next.as<uint32_t>();
  bool insertedInLds = false;
  if (generic) {
    t->mode = JMODE_HASH;
    const uint64_t cap = std::max<uint64_t>(2048, nextPow2(static_cast<uint64_t>(h.numRows) * 2));
    t->capacity = cap;
    t->gslots.ensure(static_cast<size_t>(cap) * 8 + 64);
    HIP_OK(hipMemsetAsync(t->gslots.ptr(), 0, static_cast<size_t>(cap) * 8, rt.stream));
    VX_LAUNCH("k_fill_u32", k_fill_u32, streamGrid(std::max<int64_t>(1, h.numRows), 256, 4), 256, 0,
              t->next.as<uint32_t>(), static_cast<uint64_t>(std::max<int64_t>(1, h.numRows)), kNoRow32);
    ia.gslots = t->gslots.as<uint64_t>();
  } else if (range <= arrayMax) {
    t->mode = JMODE_ARRAY;
    t->capacity = range;
    t->head.ensure(static_cast<size_t>(range) * 4 + 64);
    const uint64_t words = (range + 31) / 32;
    t->present.ensure(static_cast<size_t>(words) * 4 + 64);
    VX_LAUNCH("k_fill_u32", k_fill_u32, streamGrid(static_cast<int64_t>(words), 256, 4), 256, 0,
              t->present.as<uint32_t>(), words, 0u);
    ia.head = t->head.as<uint32_t>();
    ia.present = t->present.as<uint32_t>();
  } else {
    t->mode = JMODE_NORMALIZED;
    // Wide slots from the start (see slotAt) for an inner-join build that no cache will hold and
    // that has 8-byte dependents without nulls: VX355_JOIN_WIDE_BUILD = 0 never, 1 whenever the
    // dependents qualify, unset = builds of 2^20 rows and more.
    int wideBuild = -1;
    if (const char* e = std::getenv("VX355_JOIN_WIDE_BUILD")) {
      wideBuild = std::atoi(e);
    }
    int32_t wideFound = 0;
    if (wideBuild != 0 && h.joinType == VX355_JOIN_INNER && !h.dropDuplicates && !h.nullAsValue &&
        (wideBuild > 0 || h.numRows >= (1 << 20))) {
      for (size_t d = 0; d < h.depKinds.size() && wideFound < kWideDeps; ++d) {
        if (kindWidth(h.depKinds[d]) == 8 && !isString(h.depKinds[d]) && !((h.depNulls >> d) & 1)) {
          ia.wideDep[wideFound] = h.depVals[d].as<uint64_t>();
          t->wideDeps[wideFound++] = static_cast<int32_t>(d);
        }
      }
    }
    t->slotShift = wideFound > 0 ? 5 : 4;
    // HashTable::newHashTableEntries (HashTable.h:946-956): twice the rows, rounded up to a power of
    // two (load <= 0.5). VX355_JOIN_WIDE_TIGHT=1 sizes a wide table for the reference's upper load bound
    // (0.7, HashTable.h:143) instead: half the bytes, but 1.5 x the slots per lookup - measured on
    // config 5's shape: grouped probe 4.2 ms instead of 2.9 ms (profiles/r06_c5_variants.md).
    uint64_t cap = std::max<uint64_t>(2048, nextPow2(static_cast<uint64_t>(h.numRows) * 2));
    if (wideFound > 0 && std::getenv("VX355_JOIN_WIDE_TIGHT") && std::atoi(std::getenv("VX355_JOIN_WIDE_TIGHT")) != 0) {
      cap = std::max<uint64_t>(2048, nextPow2((static_cast<uint64_t>(h.numRows) * 10 + 6) / 7));
    }
    t->capacity = cap;
    t->slots.ensure((static_cast<size_t>(cap) << t->slotShift) + 64);
    ia.slots = t->slots.as<Slot>();
    ia.slotShift = t->slotShift;
    t->homeMask = (cap - 1) & ~static_cast<uint64_t>((64 >> t->slotShift) - 1);
    ia.homeMask = t->homeMask;
    ia.mode = t->mode;
    ia.capacity = cap;
    // Assembled group by group in LDS (k_lds_build) when the build is large: VX355_JOIN_LDS_BUILD = 0
    // never, 1 whenever possible, unset = builds of 2^20 rows and more. Not for builds that keep rows
    // with null keys out of the table (right / full joins), nor for normalized keys beyond 2^63.
    int ldsBuild = -1;
    if (const char* e = std::getenv("VX355_JOIN_LDS_BUILD")) {
      ldsBuild = std::atoi(e);
    }
    if (ldsBuild != 0 && h.numRows > 0 && ia.keyNull == nullptr && range < (1ULL << 63) &&
        (ldsBuild > 0 || h.numRows >= (1 << 20))) {
      insertedInLds = buildInLds(h, *t, ia);
    }
    if (!insertedInLds) {
      const uint64_t units = cap << (t->slotShift - 4);
      VX_LAUNCH("k_fill_slots", k_fill_slots, streamGrid(static_cast<int64_t>(units), 256, 2), 256, 0,
                t->slots.as<Slot>(), units, wideFound > 0 ? 1 : 0);
      t->wrapMask = cap - 1;
    }
    ia.wrapMask = t->wrapMask;  // (k_count_init walks the finished table)
  }
  ia.mode = t->mode;
  ia.capacity = t->capacity;
  ia.phase = 1;
  if (h.numRows > 0 && !insertedInLds) {
    VX_LAUNCH("k_join_insert", k_join_insert, streamGrid(h.numRows, 256), 256, 0, ia);
  }
  BuildCounters c = readBuildCounters(h.countersBuf);
  if (c.tableFull) {
    VX_THROW(VX355_EINTERNAL, "join table full");
  }
  if ((t->mode == JMODE_ARRAY || t->mode == JMODE_NORMALIZED) && c.duplicates && !insertedInLds) {
    ia.phase = 2;
    VX_LAUNCH("k_join_insert", k_join_insert, streamGrid(h.numRows, 256), 256, 0, ia);
    rt.sync();
  }
  t->numDistinct = c.numDistinct;
  t->hasDuplicates = c.duplicates != 0;
  if (t->slotShift == 5) {
    // the inline dependents are those of the claiming row: only meaningful when it is the key's only row
    t->wideTried = true;
    int32_t found = 0;
    while (found < kWideDeps && t->wideDeps[found] >= 0) {
      ++found;
    }
    t->numWide = t->hasDuplicates ? 0 : found;
  }
  if (countingJoin(h.joinType)) {
    const size_t bytes = static_cast<size_t>(std::max<int64_t>(1, h.numRows)) * 4;
    t->remaining.ensure(bytes + 64);
    HIP_OK(hipMemsetAsync(t->remaining.ptr(), 0, bytes, rt.stream));
    if (h.numRows > 0) {
      VX_LAUNCH("k_count_init", k_count_init, streamGrid(h.numRows, 256), 256, 0, ia, t->remaining.as<uint32_t>());
    }
    rt.sync();
  }
  if (marksProbedRows(h.joinType)) {
    t->probed.ensure(static_cast<size_t>(std::max<int64_t>(1, h.numRows)) + 64);
    HIP_OK(hipMemsetAsync(t->probed.ptr(), 0, static_cast<size_t>(std::max<int64_t>(1, h.numRows)), rt.stream));
    rt.sync();
  }
  for (auto& b : h.strBlocks) {
    t->strBlocks.push_back(std::move(b));
  }
  h.strBlocks.clear();
  for (int32_t i = 0; i < numOthers; ++i) {
    for (auto& b : others[i]->strBlocks) {
      t->strBlocks.push_back(std::move(b));
    }
    others[i]->strBlocks.clear();
  }
  t->depVals = std::move(h.depVals);
  t->depValid = std::move(h.depValid);
  t->depNulls = h.depNulls;
  // The build key images stay with the table: generic-mode probes compare
  // against them, dynamic filters (value lists, Bloom blocks) are made from them.
  t->keyStore = std::move(h.keyVals);
  h.keyVals.clear();
  if (h.nullAware && h.hasNullKeys && h.numRows > 0) {
    // the rows a null-aware join with an extra filter tests every unmatched probe row against
    t->nullKeyRows.ensure(static_cast<size_t>(h.numRows) * 4 + 64);
    uint32_t* count = reinterpret_cast<uint32_t*>(h.countersBuf.ensure(64));
    HIP_OK(hipMemsetAsync(count, 0, 4, rt.stream));
    VX_LAUNCH("k_list_null_keys", k_list_null_keys, streamGrid(h.numRows, 256), 256, 0, h.keyNull.as<uint8_t>(),
              h.numRows, t->nullKeyRows.as<uint32_t>(), count);
    uint32_t n = 0;
    copyOut(&n, VX355_MEM_HOST, count, 4);
    t->numNullKeyRows = n;
  }
  if (h.nullAsValue) {
    t->keyNull = std::move(h.keyNull);  // generic-mode probes compare null masks too
  }
  h.finished = true;
  return t.release();
}

// The filter of 'p' over the batch in 'db' and the table's dependent columns.
void fillFilterArgs(const vx355_join_probe& p, const DeviceBatch& db, JoinFilterArgs* f) {
  const auto& t = *p.table;
  *f = JoinFilterArgs{};
  f->numTerms = static_cast<int32_t>(p.filter.size());
  for (size_t d = 0; d < t.depKinds.size(); ++d) {
    f->depVals[d] = t.depVals[d].as<char>();
    f->depValid[d] = t.depValid[d].as<uint8_t>();
    f->depWidth[d] = kindWidth(t.depKinds[d]);
    f->depKind[d] = t.depKinds[d];
  }
  auto classOf = [](int32_t kind) { return isString(kind) ? 2 : ((kind == VX355_REAL || kind == VX355_DOUBLE) ? 1 : 0); };
  for (int32_t i = 0; i < f->numTerms; ++i) {
    const vx355_join_filter_term& src = p.filter[i];
    JoinFilterTerm& term = f->terms[i];
    term.leftSide = src.left_side;
    term.cmp = src.cmp;
    term.rightKind = src.right_kind;
    term.constKind = src.const_kind;
    term.i64 = src.i64;
    term.f64 = src.f64;
    int leftClass, rightClass;
    auto depKind = [&](int32_t dep) {
      VX_CHECK_ARG(dep >= 0 && dep < static_cast<int32_t>(t.depKinds.size()), "join filter: no such build column");
      return t.depKinds[dep];
    };
    auto probeCol = [&](int32_t col) -> const ColView& {
      VX_CHECK_ARG(col >= 0 && col < db.numCols() && db.used(col), "join filter: no such probe column");
      return db.col(col);
    };
    if (src.left_side == 0) {
      term.leftProbe = probeCol(src.left_col);
      leftClass = classOf(term.leftProbe.kind);
      if (term.leftProbe.kind == VX355_TIMESTAMP) {
        VX_THROW(VX355_EUNSUPPORTED, "join filter over TIMESTAMP");
      }
    } else {
      term.leftDep = src.left_col;
      leftClass = classOf(depKind(src.left_col));
      if (depKind(src.left_col) == VX355_TIMESTAMP) {
        VX_THROW(VX355_EUNSUPPORTED, "join filter over TIMESTAMP");
      }
    }
    if (src.right_kind == 1) {
      term.rightProbe = probeCol(src.right_col);
      rightClass = classOf(term.rightProbe.kind);
      if (term.rightProbe.kind == VX355_TIMESTAMP) {
        VX_THROW(VX355_EUNSUPPORTED, "join filter over TIMESTAMP");
      }
    } else if (src.right_kind == 2) {
      term.rightDep = src.right_col;
      rightClass = classOf(depKind(src.right_col));
      if (depKind(src.right_col) == VX355_TIMESTAMP) {
        VX_THROW(VX355_EUNSUPPORTED, "join filter over TIMESTAMP");
      }
    } else {
      rightClass = src.const_kind == VX355_BIGINT ? 0 : (src.const_kind == VX355_DOUBLE ? 1 : 2);
      if (rightClass == 2) {
        VX_CHECK_ARG(src.str_size >= 0 && src.str_size <= 12, "join filter: string constants of at most 12 bytes");
        char view[16] = {0};
        const uint32_t size = static_cast<uint32_t>(src.str_size);
        std::memcpy(view, &size, 4);
        std::memcpy(view + 4, src.str, size);
        std::memcpy(&term.str, view, 16);
      }
    }
    if ((leftClass == 2) != (rightClass == 2)) {
      VX_THROW(VX355_EINVAL, "join filter compares a string with a number");
    }
    if (leftClass == 2 && src.cmp != VX355_CMP_EQ && src.cmp != VX355_CMP_NE) {
      VX_THROW(VX355_EUNSUPPORTED, "join filter: strings compare with = and <> only");
    }
  }
}

int32_t ensureWide(vx355_join_table& t, int32_t mode, int64_t probeRows);

// Where the slices of a regrouped batch begin (probeAddInputRegrouped).
struct GroupedInput {
  const uint64_t* offsets;  // bin-major scan of the (bin, tile) histogram: bin b starts at offsets[b * histTiles]
  int64_t histTiles;
  int32_t numBins;
};

void probeAddInput(vx355_join_probe& p, const vx355_batch* batch, const GroupedInput* grouped = nullptr) {
  auto& rt = Runtime::get();
  auto& t = *p.table;
  DeviceBatch& db = p.batch;
  db.load(batch, p.usedCols);
  const int64_t n = db.numRows();
  const bool filtered = !p.filter.empty();
  const bool counting = countingJoin(p.joinType);
  p.numRows = n;
  p.cursor = 0;
  p.totalOut = 0;
  p.numTiles = 0;
  p.hasInput = true;
  p.haveUnitSums = false;
  if (n == 0) {
    return;
  }
  ProbeArgs a{};
  if (p.nullAware && filtered) {
    // three-valued IN over the FILTERED build rows: resolved behind k_join_filter (k_na_resolve)
    a.nullAware = 1;
    p.nullKeyValue = -2;
    p.missValue = -1;
  } else if (p.nullAware && p.joinType == VX355_JOIN_ANTI) {
    if (t.hasNullKeys) {
      return;  // NOT IN over a set holding a null: no row qualifies (HashBuild's antiJoinHasNullKeys)
    }
    a.nullAware = t.numRows > 0 ? 1 : 0;  // an empty build side passes every row, null keys included
  }
  if (p.nullAware && !filtered && p.joinType == VX355_JOIN_LEFT_SEMI_PROJECT) {
    // x IN (subquery) as a column (HashProbe.cpp:923-966, no filter): NULL for a null probe key
    // unless the build side is empty and null-free, NULL instead of FALSE once the build side
    // holds a null key
    a.nullAware = 1;
    p.nullKeyValue = (t.numRows > 0 || t.hasNullKeys) ? -2 : -1;
    p.missValue = t.hasNullKeys ? -2 : -1;
  }
  // with a filter the probed flags belong to the passing pairs: k_join_filter sets them
  a.probed = (marksProbedRows(p.joinType) && !filtered) ? t.probed.as<uint8_t>() : nullptr;
  a.numKeys = static_cast<int32_t>(p.keyCols.size());
  for (int k = 0; k < a.numKeys; ++k) {
    a.keys[k] = db.col(p.keyCols[k]);
    const int32_t kind = a.keys[k].kind;
    if (t.mode == JMODE_HASH ? kind != t.keyKinds[k]
                             : (isString(kind) != isString(t.keyKinds[k]) ||
                                (!isString(kind) && !isIntLike(kind)))) {
      VX_THROW(VX355_EUNSUPPORTED, "probe key type does not match the build key");
    }
    a.ranges[k] = t.ranges[k];
    if (t.mode == JMODE_HASH) {
      a.keyStore[k] = t.keyStore[k].as<uint64_t>();
      a.keyWords[k] = keyWordsOf(t.keyKinds[k]);
    }
  }
  a.rf.numTerms = static_cast<int32_t>(p.inputFilter.size());
  if (a.rf.numTerms > 0) {
    makeTermArgs(db, p.inputFilter.data(), a.rf.numTerms, a.rf.terms);
    const TermArg& t0 = a.rf.terms[0];
    if (a.rf.numTerms == 1 && t0.constKind == VX355_BIGINT && t0.col.enc == VX355_FLAT && t0.col.nulls == nullptr &&
        (t0.col.kind == VX355_INTEGER || t0.col.kind == VX355_BIGINT)) {
      const int64_t c = t0.i64;
      int64_t lo = INT64_MIN, hi = INT64_MAX;
      bool empty = false;
      switch (t0.cmp) {
        case VX355_CMP_EQ:
        case VX355_CMP_NE:
          lo = hi = c;
          break;
        case VX355_CMP_LT:
          empty = c == INT64_MIN;
          hi = c - (empty ? 0 : 1);
          break;
        case VX355_CMP_LE:
          hi = c;
          break;
        case VX355_CMP_GT:
          empty = c == INT64_MAX;
          lo = c + (empty ? 0 : 1);
          break;
        default:
          lo = c;
          break;
      }
      if (empty) {
        lo = 1;
        hi = 0;
      }
      a.rf.fast = t0.col.kind == VX355_INTEGER ? 4 : 8;
      a.rf.fastValues = t0.col.values;
      if (a.rf.fast == 4) {
        // the kernel compares 32-bit words: the range clamped to int32 (empty stays empty)
        if (lo > INT32_MAX || hi < INT32_MIN) {
          lo = 1;
          hi = 0;
        } else {
          lo = std::max<int64_t>(lo, INT32_MIN);
          hi = std::min<int64_t>(hi, INT32_MAX);
        }
      }
      a.rf.lo = lo;
      a.rf.hi = hi;
      a.rf.invert = t0.cmp == VX355_CMP_NE ? 1 : 0;
    }
  }
  a.mode = t.mode;
  a.nullAsValue = t.nullAsValue ? 1 : 0;
  a.keyNullStore = (t.nullAsValue && t.keepsNullRows) ? t.keyNull.as<uint8_t>() : nullptr;
  a.hasDuplicates = t.hasDuplicates ? 1 : 0;
  a.joinType = p.joinType;
  a.numRows = n;
  a.head = t.head.as<uint32_t>();
  a.slots = t.slots.as<Slot>();
  a.slotShift = t.slotShift;
  a.homeMask = t.homeMask;
  a.wrapMask = t.wrapMask;
  a.gslots = t.gslots.as<uint64_t>();
  a.capacity = t.capacity;
  a.next = t.next.as<uint32_t>();
  a.present = t.present.as<uint32_t>();
  a.hits = static_cast<uint32_t*>(p.hits.ensure(static_cast<size_t>(n) * 4 + 64));
  // Per-row output counts are only materialised when chains can be longer than
  // one row; otherwise listJoinResults derives them from the hit.
  p.haveCounts = t.hasDuplicates &&
      (p.joinType == VX355_JOIN_INNER || p.joinType == VX355_JOIN_LEFT || p.joinType == VX355_JOIN_RIGHT ||
       p.joinType == VX355_JOIN_FULL);
  a.counts = p.haveCounts ? static_cast<uint32_t*>(p.counts.ensure(static_cast<size_t>(n) * 4 + 64))
                          : nullptr;
  if (filtered) {
    // the filter pass counts the passing pairs itself
    p.haveCounts = true;
    p.counts.ensure(static_cast<size_t>(n) * 4 + 64);
    a.counts = nullptr;
  }
  p.numTiles = ceilDiv(n, kTileRows);
  a.numTiles = p.numTiles;
  uint64_t* sums = static_cast<uint64_t*>(p.tileSums.ensure(static_cast<size_t>(p.numTiles) * 8 + 64));
  uint64_t* offs =
      static_cast<uint64_t*>(p.tileOffsets.ensure(static_cast<size_t>(p.numTiles + 1) * 8 + 64));
  a.tileSums = sums;
  a.fastKey = 0;
  if (t.mode != JMODE_HASH && a.numKeys == 1 && a.keys[0].kind == VX355_BIGINT && a.keys[0].nulls == nullptr &&
      a.ranges[0].multiplier == 1) {
    a.fastKey = a.keys[0].enc == VX355_FLAT ? 1 : (a.keys[0].enc == VX355_DICTIONARY ? 2 : 0);
  }
  a.window = p.window ? 1 : 0;
  a.presentWords = (t.capacity + 31) / 32;
  // inline dependents: inner joins whose output is one row per hit (no filter, no counting)
  p.wideStaged = 0;
  // (not with a fused input filter: the WIDE instantiations are built without it)
  if (a.fastKey == 1 && p.joinType == VX355_JOIN_INNER && !filtered && !counting && !a.nullAware && p.inputFilter.empty()) {
    p.wideStaged = ensureWide(t, p.wideMode, n);
    if (p.wideStaged > 0) {
      a.wide = t.slotShift == 5 ? t.slots.as<WideSlot>() : t.wide.as<WideSlot>();
      for (int i = 0; i < p.wideStaged; ++i) {
        a.hitVals[i] = static_cast<uint64_t*>(p.hitVals[i].ensure(static_cast<size_t>(n) * 8 + 64));
      }
    }
  }
  auto launch = [&](auto sparseTag, int64_t tileBegin, int64_t tileEnd) {
    constexpr bool SP = decltype(sparseTag)::value;
    ProbeArgs la = a;
    la.tileBegin = tileBegin;
    la.numTiles = tileEnd;
    la.countHits = tileBegin == 0 && tileEnd < p.numTiles ? 1 : 0;
    const int grid =
        static_cast<int>(std::min<int64_t>(tileEnd - tileBegin, static_cast<int64_t>(rt.numCUs) * 8));
    if (grid <= 0) {
      return;
    }
    // profile names: "k_join_probe" writes hits[] for k_emit, "k_join_probe_list" also lists the hits
    const char* name = SP ? "k_join_probe_list" : "k_join_probe";
    if (la.rf.numTerms > 0 && !(a.fastKey == 1 && la.rf.fast && la.wide == nullptr && t.mode != JMODE_HASH)) {
      // any other filter shape (several terms, nulls, dictionaries, strings, doubles) or key shape: the
      // all-purpose probe with the generic evaluator
      VX_LAUNCH(name, (k_join_probe<-1, -1, SP, 0, -1>), grid, 256, 0, la);
    } else if (t.mode == JMODE_ARRAY && a.fastKey == 1 && la.rf.fast == 4) {
      VX_LAUNCH(name, (k_join_probe<JMODE_ARRAY, 1, SP, 0, 4>), grid, 256, 0, la);
    } else if (t.mode == JMODE_ARRAY && a.fastKey == 1 && la.rf.fast == 8) {
      VX_LAUNCH(name, (k_join_probe<JMODE_ARRAY, 1, SP, 0, 8>), grid, 256, 0, la);
    } else if (t.mode == JMODE_NORMALIZED && a.fastKey == 1 && la.rf.fast == 4 && la.wide == nullptr) {
      VX_LAUNCH(name, (k_join_probe<JMODE_NORMALIZED, 1, SP, 0, 4>), grid, 256, 0, la);
    } else if (t.mode == JMODE_NORMALIZED && a.fastKey == 1 && la.rf.fast == 8 && la.wide == nullptr) {
      VX_LAUNCH(name, (k_join_probe<JMODE_NORMALIZED, 1, SP, 0, 8>), grid, 256, 0, la);
    } else if (t.mode == JMODE_ARRAY && a.fastKey == 1) {
      VX_LAUNCH(name, (k_join_probe<JMODE_ARRAY, 1, SP>), grid, 256, 0, la);
    } else if (t.mode == JMODE_ARRAY && a.fastKey == 2) {
      VX_LAUNCH(name, (k_join_probe<JMODE_ARRAY, 2, SP>), grid, 256, 0, la);
    } else if (t.mode == JMODE_NORMALIZED && a.fastKey == 1 && la.wide != nullptr) {
      if (la.hitVals[1] != nullptr) {
        VX_LAUNCH(name, (k_join_probe<JMODE_NORMALIZED, 1, SP, 2>), grid, 256, 0, la);
      } else {
        VX_LAUNCH(name, (k_join_probe<JMODE_NORMALIZED, 1, SP, 1>), grid, 256, 0, la);
      }
    } else if (t.mode == JMODE_NORMALIZED && a.fastKey == 1) {
      VX_LAUNCH(name, (k_join_probe<JMODE_NORMALIZED, 1, SP>), grid, 256, 0, la);
    } else if (t.mode == JMODE_ARRAY) {
      VX_LAUNCH(name, (k_join_probe<JMODE_ARRAY, 0, SP>), grid, 256, 0, la);
    } else {
      VX_LAUNCH(name, (k_join_probe<-1, -1, SP>), grid, 256, 0, la);
    }
  };
  // Sparse listing: joins that emit matches only, one build row per match, not null
  // aware. Whether the hit rate is low enough is measured on the first tiles of the
  // batch (the probe pass reports tiles that overflowed their staging segment).
  // (a batch whose sample sent it to the dense form lets the next seven batches of this operator
  // skip the sample: the hit rate of a stream does not change from batch to batch)
  const bool recentlyDense = p.sparseMode < 0 && p.denseStreak > 0;
  if (recentlyDense) {
    --p.denseStreak;
  }
  const bool sparseEligible = p.sparseMode != 0 && !recentlyDense && !p.haveCounts && !a.nullAware && !filtered &&
      !counting &&
      (p.joinType == VX355_JOIN_INNER || p.joinType == VX355_JOIN_LEFT_SEMI_FILTER ||
       (p.joinType == VX355_JOIN_RIGHT && !t.hasDuplicates));
  p.sparse = false;
  p.pairList = false;
  p.sparseTiles = p.denseTiles = 0;
  // Range-partitioned probe (see k_pp_*): flat BIGINT key into an array-mode table whose bitmap
  // is far larger than an L2, a batch big enough to pay for two extra passes.
  const uint64_t presentWords = (t.capacity + 31) / 32;
  const int64_t partBins = static_cast<int64_t>((t.capacity + (1ULL << kPartShift) - 1) >> kPartShift);
  const bool partEligible = p.partitionMode != 0 && t.mode == JMODE_ARRAY && a.fastKey == 1 &&
      partBins <= kPartMaxBins && n < (1LL << 32) &&
      (p.partitionMode == 1 || (n >= (4LL << 20) && presentWords * 4 > (16ULL << 20)));
  if (grouped && t.mode == JMODE_NORMALIZED && a.rf.numTerms == 0) {
    // the batch arrives grouped by slice of the slot array: slice by slice, one XCD per slice
    const int unitRows = p.groupUnit;
    const int64_t units = ceilDiv(n, unitRows);
    int4* unitList = static_cast<int4*>(p.grpUnits.ensure(static_cast<size_t>(units) * sizeof(int4) + 64));
    uint32_t* xcdStart = static_cast<uint32_t*>(p.grpXcd.ensure(64));
    VX_LAUNCH("k_grp_units", k_grp_units, static_cast<int>(ceilDiv(units, 1024)), 1024, 0, grouped->offsets,
              grouped->histTiles, grouped->numBins, units, unitRows, unitList, xcdStart);
    HIP_OK(hipMemsetAsync(sums, 0, static_cast<size_t>(p.numTiles) * 8, rt.stream));
    // (not when a later pass rewrites hits[] - counting joins, extra filters - or per-row counts exist)
    if (unitRows == kTileRows / 4 && !filtered && !counting && !p.haveCounts) {
      a.unitSums = static_cast<uint32_t*>(p.grpUnitSums.ensure(static_cast<size_t>(p.numTiles) * 4 * 4 + 64));
      HIP_OK(hipMemsetAsync(a.unitSums, 0, static_cast<size_t>(p.numTiles) * 4 * 4, rt.stream));  // (quarters past the last row)
      p.haveUnitSums = true;
    }
    // workgroups per CU such that the rows in flight on an XCD (CUs / 8 x workgroups x unit) are about
    // 1.25 slices' worth, between 2 and 8; the launch's unused LDS request enforces it
    const int64_t cusPerXcd = std::max(1, rt.numCUs / 8);
    int64_t perCu = p.groupWgs > 0 ? p.groupWgs : ceilDiv(n / grouped->numBins * 5 / 4, cusPerXcd * unitRows);
    perCu = std::min<int64_t>(std::max<int64_t>(perCu, 2), 8);
    const size_t ldsPad = perCu >= 8 ? 0 : (static_cast<size_t>(160) << 10) / static_cast<size_t>(perCu) - 1024;
    const int grid = static_cast<int>(ceilDiv(units, 8) * 8);
    const int shift = a.wide != nullptr ? 5 : t.slotShift;
    const uint4* tableBase = a.wide != nullptr ? reinterpret_cast<const uint4*>(a.wide) : reinterpret_cast<const uint4*>(a.slots);
    const uint64_t sliceUnits16 = p.groupPrefetch ? ((t.capacity / static_cast<uint64_t>(grouped->numBins)) << shift) / 16 : 0;
    auto launchGrouped = [&](auto kernel) {
      if (ldsPad > (48u << 10)) {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(ldsPad)));
      }
      VX_LAUNCH("k_join_probe_grouped", kernel, grid, 256, ldsPad, a, unitList, xcdStart, tableBase, sliceUnits16);
    };
    const int wideKind = a.fastKey == 1 ? (a.wide != nullptr ? (a.hitVals[1] != nullptr ? 2 : 1) : 0) : -1;
    if (unitRows == 8192) {
      switch (wideKind) {
        case 2: launchGrouped(k_join_probe_grouped<1, 2, 8192>); break;
        case 1: launchGrouped(k_join_probe_grouped<1, 1, 8192>); break;
        case 0: launchGrouped(k_join_probe_grouped<1, 0, 8192>); break;
        default: launchGrouped(k_join_probe_grouped<-1, 0, 8192>); break;
      }
    } else {
      switch (wideKind) {
        case 2: launchGrouped(k_join_probe_grouped<1, 2, 2048>); break;
        case 1: launchGrouped(k_join_probe_grouped<1, 1, 2048>); break;
        case 0: launchGrouped(k_join_probe_grouped<1, 0, 2048>); break;
        default: launchGrouped(k_join_probe_grouped<-1, 0, 2048>); break;
      }
    }
  } else if (sparseEligible) {
    a.staged = static_cast<uint2*>(p.staged.ensure(static_cast<size_t>(p.numTiles) * kSparseCap * sizeof(uint2) + 64));
    a.tileDense = static_cast<uint8_t*>(p.tileDense.ensure(static_cast<size_t>(p.numTiles) + 64));
    a.sparseStats = static_cast<uint64_t*>(p.sparseStats.ensure(64));
    HIP_OK(hipMemsetAsync(a.sparseStats, 0, 32, rt.stream));
    if (partEligible) {
      VX_LAUNCH("k_key_locality", k_key_locality, 64, 64, 0, static_cast<const int64_t*>(a.keys[0].values), n,
                reinterpret_cast<uint32_t*>(a.sparseStats + 2));
    }
    const bool decideHere = partEligible || p.sparseMode != 1;
    const int64_t sample = decideHere ? std::min<int64_t>(p.numTiles, 256) : p.numTiles;
    launch(std::true_type{}, 0, sample);
    p.sparse = true;
    if (sample < p.numTiles || partEligible) {
      uint64_t stats[4] = {0, 0, 0, 0};
      copyOut(stats, VX355_MEM_HOST, a.sparseStats, 32);  // synchronises the stream
      const uint64_t overflowed = stats[0];
      // dense when more than 1 tile in 16 overflowed or the average tile is 7/8 full. (Until round 5: half
      // full. TPC-H Q3's orders probe with its date filter fused in hits on 9.7 % of its rows - 800 of the
      // 1024 staged pairs a tile holds, 200 +- 13 of a wave's 256: no overflows - and listing them in the
      // probe pass costs 117 MB of pairs instead of 600 MB of hits[] written and read again by k_emit.)
      const bool stay = p.sparseMode == 1 ||
          (overflowed * 16 <= static_cast<uint64_t>(sample) &&
           stats[1] <= static_cast<uint64_t>(sample) * (kSparseCap - kSparseCap / 8));
      // scattered probe keys: more than 16 distinct bitmap lines per wave on average
      const bool scattered = p.partitionMode == 1 || (stats[2] & 0xffffffffULL) > 64ULL * 16;
      if (stay && partEligible && scattered) {
        p.sparse = false;
        p.pairList = true;
        PartArgs pa{};
        pa.keys = static_cast<const int64_t*>(a.keys[0].values);
        pa.numRows = n;
        pa.numTiles = ceilDiv(n, kPartTileRows);
        pa.keyMin = a.ranges[0].min;
        pa.keyMax = a.ranges[0].max;
        pa.numBins = static_cast<int32_t>(partBins);
        pa.rf = a.rf;
        const int64_t cells = pa.numTiles * pa.numBins;
        const int pgrid = static_cast<int>(std::min<int64_t>(pa.numTiles, static_cast<int64_t>(rt.numCUs) * 2));
        PartProbeArgs pp{};
        bool counted = true;
        if (p.partitionFast && pa.numBins <= kPartFastBins) {
          PartFastArgs fa{};
          fa.keys = pa.keys;
          fa.numRows = n;
          fa.keyMin = pa.keyMin;
          fa.keyMax = pa.keyMax;
          fa.numBins = pa.numBins;
          fa.rf = a.rf;
          fa.binCap = static_cast<uint64_t>(n / pa.numBins + n / (2 * pa.numBins) + 4096);
          fa.recs = static_cast<uint64_t*>(p.ppRecs.ensure(static_cast<size_t>(fa.binCap) * pa.numBins * 8 + 64));
          fa.binCount = static_cast<uint32_t*>(p.ppHist.ensure(static_cast<size_t>(pa.numBins + 16) * 4 + 64));
          fa.overflow = fa.binCount + pa.numBins;
          HIP_OK(hipMemsetAsync(fa.binCount, 0, static_cast<size_t>(pa.numBins + 16) * 4, rt.stream));
          const int64_t numSub = ceilDiv(n, kPartFastSub);
          VX_LAUNCH("k_pp_scatter", k_pp_scatter_fast, static_cast<int>(std::min<int64_t>(numSub, rt.numCUs * 2)), 1024, 0, fa);
          uint32_t full = 0;
          copyOut(&full, VX355_MEM_HOST, fa.overflow, 4);
          counted = full != 0;   // a bin outgrew its region (skewed keys): the counted passes below
          if (!counted) {
            pp.recs = fa.recs;
            pp.binCount = fa.binCount;
            pp.binCap = fa.binCap;
          } else {
            p.partitionFast = false;
          }
        }
        uint64_t* offsets = nullptr;
        if (counted) {
          pa.hist = static_cast<uint32_t*>(p.ppHist.ensure(static_cast<size_t>(cells) * 4 + 64));
          offsets = static_cast<uint64_t*>(p.ppOffsets.ensure(static_cast<size_t>(cells + 1) * 8 + 64));
          pa.offsets = offsets;
          pa.recs = static_cast<uint64_t*>(p.ppRecs.ensure(static_cast<size_t>(n) * 8 + 64));
          VX_LAUNCH("k_pp_count", k_pp_count, pgrid, 1024, 0, pa);
          scanU32ToU64(pa.hist, cells, offsets, p.ppScan);
          VX_LAUNCH("k_pp_scatter", k_pp_scatter, pgrid, 1024, 0, pa);
          pp.recs = pa.recs;
        }
        pp.offsets = offsets;
        pp.numTiles = pa.numTiles;
        pp.numBins = pa.numBins;
        pp.present = a.present;
        pp.presentWords = presentWords;
        pp.head = a.head;
        pp.probed = a.probed;
        // at most one hit per probe row (chains of one row, or the head only for semi joins)
        pp.pairs = static_cast<uint64_t*>(p.ppPairs.ensure(static_cast<size_t>(n) * 8 + 64));
        pp.numPairs = reinterpret_cast<unsigned long long*>(a.sparseStats + 3);
        VX_LAUNCH("k_join_probe_part", k_pp_probe, std::min<int>(pa.numBins, rt.numCUs * 2), 1024, 0, pp);
        uint64_t numPairs = 0;
        copyOut(&numPairs, VX355_MEM_HOST, a.sparseStats + 3, 8);
        if (numPairs > 0) {
          // {probe row : 32 | build row : 32}, at most one hit per probe row: the probe-row bits decide
          const int rowBits = std::max(1, 64 - __builtin_clzll(static_cast<unsigned long long>(std::max<int64_t>(1, n))));
          sortKeysU64(pp.pairs, static_cast<uint64_t*>(p.ppSorted.ensure(static_cast<size_t>(numPairs) * 8 + 64)),
                      static_cast<size_t>(numPairs), p.ppSortTmp, 32, 32 + rowBits);
        }
        rt.sync();
        p.totalOut = numPairs;
        return;
      }
      if (sample >= p.numTiles) {
        // whole batch already probed by the sample launch
      } else if (stay) {
        launch(std::true_type{}, sample, p.numTiles);
      } else {
        HIP_OK(hipMemsetAsync(a.tileDense + sample, 1, static_cast<size_t>(p.numTiles - sample), rt.stream));
        launch(std::false_type{}, sample, p.numTiles);
        p.denseTiles = p.numTiles - sample;
        p.denseStreak = 7;
      }
    }
  } else {
    launch(std::false_type{}, 0, p.numTiles);
  }
  if (counting && t.numRows > 0) {
    // Rank the probe rows of every key in probe order and let the first 'remaining' consume.
    const size_t words = static_cast<size_t>(ceilDiv(n, 64));
    uint64_t* bits = static_cast<uint64_t*>(p.hitBits.ensure(words * 8 + 64));
    VX_LAUNCH("k_hit_bits", k_hit_bits, streamGrid(n, 256), 256, 0, a.hits, n, bits);
    int32_t* rows = static_cast<int32_t*>(p.hitRows.ensure(static_cast<size_t>(n) * 4 + 64));
    int64_t numHits = 0;
    compactBits(bits, nullptr, nullptr, n, rows, p.scratch, &numHits);
    if (numHits > 0) {
      uint64_t* wordsIn = static_cast<uint64_t*>(p.hitWords.ensure(static_cast<size_t>(numHits) * 8 + 64));
      uint64_t* sorted = static_cast<uint64_t*>(p.hitSorted.ensure(static_cast<size_t>(numHits) * 8 + 64));
      VX_LAUNCH("k_hit_words", k_hit_words, streamGrid(numHits, 256), 256, 0, a.hits, rows, numHits, wordsIn);
      // {head build row : 32 | probe row : 32}; the words arrive in ascending probe-row order (compactBits),
      // so a STABLE sort on the build-row bits alone gives (build row, probe row) order
      const int headBits = std::max(1, 64 - __builtin_clzll(static_cast<unsigned long long>(std::max<int64_t>(1, t.numRows))));
      sortKeysU64(wordsIn, sorted, static_cast<size_t>(numHits), p.sortTmp, 32, 32 + headBits);
      VX_LAUNCH("k_count_consume", k_count_consume, streamGrid(numHits, 256), 256, 0, sorted, numHits,
                t.remaining.as<uint32_t>(), a.hits);
      VX_LAUNCH("k_count_settle", k_count_settle, streamGrid(numHits, 256), 256, 0, sorted, numHits,
                t.remaining.as<uint32_t>());
    }
  }
  if (filtered || counting) {
    FilterPassArgs fa{};
    fillFilterArgs(p, db, &fa.f);
    fa.hits = a.hits;
    fa.counts = static_cast<uint32_t*>(p.counts.ensure(static_cast<size_t>(n) * 4 + 64));
    p.haveCounts = true;
    fa.tileSums = sums;
    fa.next = a.next;
    // a counting join asks about the head row only (one row per distinct key consumed)
    fa.probed = (filtered && marksProbedRows(p.joinType)) ? t.probed.as<uint8_t>() : nullptr;
    fa.numRows = n;
    fa.numTiles = p.numTiles;
    fa.joinType = p.joinType;
    const int fgrid = static_cast<int>(std::min<int64_t>(p.numTiles, static_cast<int64_t>(rt.numCUs) * 8));
    VX_LAUNCH("k_join_filter", k_join_filter, fgrid, 256, 0, fa);
    if (filtered && p.nullAware) {
      // HashProbe::evalFilterForNullAwareJoin: rows without a passing equal-key pair meet the
      // null-key build rows (key not null) or every build row (key null)
      uint32_t* pending = static_cast<uint32_t*>(p.hitRows.ensure(static_cast<size_t>(n) * 4 + 64));
      uint32_t* count = reinterpret_cast<uint32_t*>(p.sparseStats.ensure(64));
      HIP_OK(hipMemsetAsync(count, 0, 4, rt.stream));
      VX_LAUNCH("k_na_pending", k_na_pending, streamGrid(n, 256), 256, 0, a.hits, n, t.numNullKeyRows, pending, count);
      uint32_t numPending = 0;
      copyOut(&numPending, VX355_MEM_HOST, count, 4);
      if (numPending > 0) {
        NullAwareArgs na{};
        na.f = fa.f;
        na.hits = a.hits;
        na.pending = pending;
        na.numPending = numPending;
        na.nullKeyRows = t.nullKeyRows.as<uint32_t>();
        na.numNullKeyRows = t.numNullKeyRows;
        na.numBuildRows = t.numRows;
        const int64_t waves = std::min<int64_t>(numPending, static_cast<int64_t>(rt.numCUs) * 32);
        VX_LAUNCH("k_na_resolve", k_na_resolve, static_cast<int>(ceilDiv(waves, 4)), 256, 0, na);
      }
      VX_LAUNCH("k_na_counts", k_na_counts, fgrid, 256, 0, a.hits, n, p.numTiles, p.joinType, fa.counts, sums);
    }
  }
  VX_LAUNCH("k_scan_u64", k_scan_u64, 1, 1024, 0, sums, p.numTiles, offs);
  p.hostTileOffsets.resize(p.numTiles + 1);
  copyOut(p.hostTileOffsets.data(), VX355_MEM_HOST, offs, static_cast<size_t>(p.numTiles + 1) * 8);
  rt.sync();
  p.totalOut = p.hostTileOffsets.back();
  if (p.sparse) {
    uint64_t stats[2] = {0, 0};
    copyOut(stats, VX355_MEM_HOST, p.sparseStats.ptr(), 16);
    p.denseTiles += static_cast<int64_t>(stats[0]);
    p.sparseTiles = p.numTiles - p.denseTiles;
  }
}

// HashProbe::addInput with the batch regrouped by slice of the table's slot array first (see
// k_grp_hist). *regrouped = 0: the table or the batch does not qualify and the batch was probed as
// it came (outValues untouched).
void probeAddInputRegrouped(vx355_join_probe& p, const vx355_batch* batch, void* const* outValues, int32_t* regrouped) {
  auto& rt = Runtime::get();
  auto& t = *p.table;
  *regrouped = 0;
  const int64_t n = batch->num_rows;
  const uint64_t tableBytes = t.capacity << t.slotShift;
  bool ok = p.regroupMode != 0 && t.mode == JMODE_NORMALIZED && t.numRows > 0 && n > 0 &&
      batch->num_cols <= kGrpMaxCols && p.inputFilter.empty();
  // worth two more passes over the batch: a table that no cache holds, a batch that fills the chip
  ok = ok && (p.regroupMode == 1 || (tableBytes >= (64ULL << 20) && n >= (1 << 22)));
  for (int32_t c = 0; ok && c < batch->num_cols; ++c) {
    const vx355_column& col = batch->cols[c];
    const int w = kindWidth(col.type_kind);
    ok = col.encoding == VX355_FLAT && col.nulls == nullptr && col.values != nullptr &&
        (w == 1 || w == 2 || w == 4 || w == 8 || w == 16) && outValues[c] != nullptr;
  }
  for (size_t k = 0; ok && k < p.keyCols.size(); ++k) {
    const int32_t kind = batch->cols[p.keyCols[k]].type_kind;
    ok = isString(kind) == isString(t.keyKinds[k]) && (isString(kind) || isIntLike(kind));
  }
  if (!ok) {
    probeAddInput(p, batch);
    return;
  }
  std::vector<int32_t> all(batch->num_cols);
  for (int32_t c = 0; c < batch->num_cols; ++c) {
    all[c] = c;
  }
  DeviceBatch in;
  in.load(batch, all);  // host columns are staged; device columns aliased
  GroupArgs g{};
  g.numKeys = static_cast<int32_t>(p.keyCols.size());
  for (int k = 0; k < g.numKeys; ++k) {
    g.keys[k] = in.col(p.keyCols[k]);
    g.ranges[k] = t.ranges[k];
  }
  g.nullAsValue = t.nullAsValue ? 1 : 0;
  g.fastKey = (g.numKeys == 1 && g.keys[0].kind == VX355_BIGINT && g.ranges[0].multiplier == 1) ? 1 : 0;
  uint64_t bins = nextPow2(std::max<uint64_t>(1, tableBytes / static_cast<uint64_t>(p.sliceBytes)));
  bins = std::min<uint64_t>(std::max<uint64_t>(bins, 8), std::min<uint64_t>(kGrpMaxBins, t.capacity));
  g.numBins = static_cast<int32_t>(bins);
  g.mask = t.capacity - 1;
  g.binShift = __builtin_ctzll(t.capacity) - __builtin_ctzll(bins);
  g.numCols = batch->num_cols;
  g.numRows = n;
  g.numTiles = ceilDiv(n, kGrpTileRows);
  // (a single flat BIGINT key goes first: k_grp_scatter takes its bins from part 0's values)
  std::vector<int32_t> order;
  if (g.fastKey) {
    order.push_back(p.keyCols[0]);
  }
  for (int32_t c = 0; c < batch->num_cols; ++c) {
    if (!g.fastKey || c != p.keyCols[0]) {
      order.push_back(c);
    }
  }
  for (int32_t c : order) {
    const int w = kindWidth(batch->cols[c].type_kind);
    for (int part = 0; part < (w == 16 ? 2 : 1); ++part) {
      const int q = g.numParts++;
      g.in[q] = static_cast<const char*>(in.col(c).values) + part * 8;
      g.out[q] = static_cast<char*>(outValues[c]) + part * 8;
      g.stride[q] = w;
      g.outStride[q] = w;
      g.width[q] = w == 16 ? 8 : w;
    }
  }
  const int64_t cells = g.numTiles * g.numBins;
  g.hist = static_cast<uint32_t*>(p.grpHist.ensure(static_cast<size_t>(cells) * 4 + 64));
  uint64_t* offsets = static_cast<uint64_t*>(p.grpOffsets.ensure(static_cast<size_t>(cells + 1) * 8 + 64));
  g.offsets = offsets;
  const int grid = static_cast<int>(std::min<int64_t>(g.numTiles, static_cast<int64_t>(rt.numCUs) * 2));
  VX_LAUNCH("k_grp_hist", k_grp_hist, grid, 1024, 0, g);
  scanU32ToU64(g.hist, cells, offsets, p.grpScan);
  launchGroupScatter(g, grid);
  std::vector<vx355_column> cols(batch->cols, batch->cols + batch->num_cols);
  for (int32_t c = 0; c < batch->num_cols; ++c) {
    cols[c].values = outValues[c];
    cols[c].mem = VX355_MEM_DEVICE;
  }
  vx355_batch moved{batch->num_rows, batch->num_cols, cols.data()};
  GroupedInput gi{offsets, g.numTiles, g.numBins};
  probeAddInput(p, &moved, &gi);
  *regrouped = 1;
}

// extractColumns for 'n' listed build rows (-1 = null row) into caller columns.
// The table's wide slots, built now if this is the first batch to ask and the table qualifies:
// normalized-key mode, no duplicate keys, at least one 8-byte dependent without nulls. A batch
// qualifies when it is at least twice the build side (the pass writes capacity x 32 bytes and
// gathers one dependent per build row; it saves one random access per probe row). -> number of
// inline dependents (0 = none).
int32_t ensureWide(vx355_join_table& t, int32_t mode, int64_t probeRows) {
  if (mode == 0 || t.mode != JMODE_NORMALIZED || t.hasDuplicates || t.numRows == 0) {
    return 0;
  }
  if (t.slotShift == 5) {
    return t.numWide;  // built wide (or not usable: duplicate keys)
  }
  std::lock_guard<std::mutex> lock(t.lazyMutex);
  if (t.numWide > 0) {
    return t.numWide;
  }
  if (t.wideTried || (mode < 0 && (probeRows < (1 << 20) || probeRows < 2 * t.numRows))) {
    return 0;
  }
  t.wideTried = true;
  int32_t found = 0;
  for (size_t d = 0; d < t.depKinds.size() && found < kWideDeps; ++d) {
    if (kindWidth(t.depKinds[d]) == 8 && !isString(t.depKinds[d]) && !((t.depNulls >> d) & 1)) {
      t.wideDeps[found++] = static_cast<int32_t>(d);
    }
  }
  if (found == 0) {
    return 0;
  }
  auto& rt = Runtime::get();
  t.wide.ensure(static_cast<size_t>(t.capacity) * sizeof(WideSlot) + 64);
  VX_LAUNCH("k_widen_slots", k_widen_slots, streamGrid(static_cast<int64_t>(t.capacity), 256, 2), 256, 0,
            t.slots.as<Slot>(), t.wide.as<WideSlot>(), t.capacity, t.depVals[t.wideDeps[0]].as<uint64_t>(),
            found > 1 ? t.depVals[t.wideDeps[1]].as<uint64_t>() : nullptr);
  rt.sync();
  t.numWide = found;
  return found;
}

void gatherBuildCols(vx355_join_probe& p, const vx355_join_table& t, const int32_t* dRows, int32_t n,
                     vx355_out_column* buildCols, const int32_t* buildColIds, int32_t numBuildCols) {
    VX_CHECK_ARG(buildCols && buildColIds, "NULL build column arguments");
    const size_t words = static_cast<size_t>(ceilDiv(n, 64));
    GatherArgs ga{};
    ga.buildRows = dRows;
    ga.count = n;
    ga.numCols = numBuildCols;
    std::vector<size_t> valOff(numBuildCols), nullOff(numBuildCols), valBytes(numBuildCols);
    size_t total = 0;
    for (int32_t c = 0; c < numBuildCols; ++c) {
      const int32_t id = buildColIds[c];
      if (id == VX355_BUILD_COL_MATCH) {
        VX_CHECK_ARG(p.joinType == VX355_JOIN_RIGHT_SEMI_PROJECT && buildCols[c].type_kind == VX355_BOOLEAN,
                     "the match column is a BOOLEAN of right semi project joins");
      } else {
        VX_CHECK_ARG(id >= 0 && id < static_cast<int32_t>(t.depKinds.size()), "bad build column id");
        VX_CHECK_ARG(buildCols[c].type_kind == t.depKinds[id], "build column type mismatch");
      }
      const int w = id == VX355_BUILD_COL_MATCH ? 0 : kindWidth(t.depKinds[id]);
      valBytes[c] = w == 0 ? words * 8 : static_cast<size_t>(n) * w;
      valOff[c] = total;
      total += (valBytes[c] + 63) & ~static_cast<size_t>(63);
      nullOff[c] = total;
      total += (words * 8 + 63) & ~static_cast<size_t>(63);
    }
    char* scratch = static_cast<char*>(p.scratch.ensure(total + 64));
    for (int32_t c = 0; c < numBuildCols; ++c) {
      const int32_t id = buildColIds[c];
      if (id == VX355_BUILD_COL_MATCH) {
        // extractProbedFlags (not null aware): match = some probe row matched this build row
        ga.depVals[c] = t.probed.as<char>();
        ga.depValid[c] = nullptr;
        ga.width[c] = 0;
        ga.kind[c] = VX355_BOOLEAN;
      } else {
        ga.depVals[c] = t.depVals[id].as<char>();
        ga.depValid[c] = ((t.depNulls >> id) & 1) ? t.depValid[id].as<uint8_t>() : nullptr;
        ga.width[c] = kindWidth(t.depKinds[id]);
        ga.kind[c] = t.depKinds[id];
      }
      const bool colHost = buildCols[c].mem == VX355_MEM_HOST;
      ga.outVals[c] = colHost ? static_cast<void*>(scratch + valOff[c]) : buildCols[c].values;
      ga.outNulls[c] = colHost ? reinterpret_cast<uint64_t*>(scratch + nullOff[c]) : buildCols[c].nulls;
    }
    VX_LAUNCH("k_gather_deps", k_gather_deps, static_cast<int>(ceilDiv(n, 256)), 256, 0, ga);
    bool longStrings = false;
    for (int32_t c = 0; c < numBuildCols; ++c) {
      if (buildCols[c].mem == VX355_MEM_HOST) {
        copyOutAsync(buildCols[c].values, VX355_MEM_HOST, scratch + valOff[c], valBytes[c]);
        if (buildCols[c].nulls) {
          copyOutAsync(buildCols[c].nulls, VX355_MEM_HOST, scratch + nullOff[c], words * 8);
        }
        longStrings = longStrings || (isString(buildCols[c].type_kind) && !t.strBlocks.empty());
      }
    }
    p.hostStrings.clear();
    if (longStrings) {
      // payload strings longer than 12 bytes: the views hold pointers into the table's HBM arena; a
      // host caller gets them re-pointed into a buffer this handle keeps until its next output call
      Runtime::get().sync();
      for (int32_t c = 0; c < numBuildCols; ++c) {
        if (buildCols[c].mem == VX355_MEM_HOST && isString(buildCols[c].type_kind)) {
          fetchLongStrings(static_cast<char*>(buildCols[c].values), n, p.hostStrings);
        }
      }
    }
}

void probeGetOutput(vx355_join_probe& p, int32_t maxRows, int32_t* mappingOut, int32_t* buildRowsOut,
                    int32_t outMem, vx355_out_column* buildCols, const int32_t* buildColIds,
                    int32_t numBuildCols, int32_t* nOut, int32_t* finished) {
  auto& rt = Runtime::get();
  auto& t = *p.table;
  VX_CHECK_ARG(nOut && finished, "NULL argument");
  VX_CHECK_ARG(maxRows > 0, "max_rows must be positive");
  VX_CHECK_ARG(numBuildCols >= 0 && numBuildCols <= kMaxDeps, "bad number of build columns");
  *nOut = 0;
  *finished = 1;
  if (!p.hasInput || p.cursor >= p.totalOut) {
    return;
  }
  VX_CHECK_ARG(mappingOut != nullptr, "mapping_out is NULL");
  if (p.outputBatchBytes > 0 && numBuildCols > 0) {
    // listJoinResults stops once the projected build columns of the listed hits reach
    // preferredOutputBatchBytes (JoinResultEmitter, HashTable.cpp:2087-2108): with fixed-width
    // columns that is a row bound. Every output row is priced as a hit, so a page is never
    // larger than the reference's.
    int64_t rowBytes = 0;
    for (int32_t c = 0; c < numBuildCols; ++c) {
      const int32_t id = buildColIds ? buildColIds[c] : -1;
      rowBytes += (id >= 0 && id < static_cast<int32_t>(t.depKinds.size())) ? std::max(1, kindWidth(t.depKinds[id])) : 1;
    }
    const int64_t byBytes = std::max<int64_t>(1, ceilDiv(p.outputBatchBytes, std::max<int64_t>(1, rowBytes)));
    maxRows = static_cast<int32_t>(std::min<int64_t>(maxRows, byBytes));
  }
  const uint64_t begin = p.cursor;
  const uint64_t end = std::min<uint64_t>(p.totalOut, begin + static_cast<uint64_t>(maxRows));
  const int32_t n = static_cast<int32_t>(end - begin);
  const bool host = outMem == VX355_MEM_HOST;
  int32_t* dMap = host ? static_cast<int32_t*>(p.outMap.ensure(static_cast<size_t>(n) * 4 + 64)) : mappingOut;
  const bool needRows = buildRowsOut != nullptr || numBuildCols > 0;
  int32_t* dRows = nullptr;
  if (needRows) {
    dRows = (host || !buildRowsOut)
        ? static_cast<int32_t*>(p.outRows.ensure(static_cast<size_t>(n) * 4 + 64))
        : buildRowsOut;
  }
  if (p.pairList) {
    // range-partitioned probe: the hits already sit in probe-row order
    VX_LAUNCH("k_emit_pairs", k_emit_pairs, static_cast<int>(ceilDiv(n, 256)), 256, 0, p.ppSorted.as<uint64_t>(),
              static_cast<int64_t>(begin), n, listsMatches(p.joinType) ? 1 : 0, dMap, dRows);
    if (numBuildCols > 0) {
      gatherBuildCols(p, t, dRows, n, buildCols, buildColIds, numBuildCols);
    }
    if (host) {
      copyOutAsync(mappingOut, VX355_MEM_HOST, dMap, static_cast<size_t>(n) * 4);
      if (buildRowsOut) {
        copyOutAsync(buildRowsOut, VX355_MEM_HOST, dRows, static_cast<size_t>(n) * 4);
      }
    }
    rt.sync();
    p.cursor = end;
    *nOut = n;
    *finished = p.cursor >= p.totalOut ? 1 : 0;
    return;
  }
  // Tiles whose offset range intersects the window.
  const auto& to = p.hostTileOffsets;
  int64_t firstTile = std::upper_bound(to.begin(), to.end(), begin) - to.begin() - 1;
  int64_t lastTile = std::lower_bound(to.begin(), to.end(), end) - to.begin() - 1;
  firstTile = std::max<int64_t>(0, std::min<int64_t>(firstTile, p.numTiles - 1));
  lastTile = std::max<int64_t>(firstTile, std::min<int64_t>(lastTile, p.numTiles - 1));
  EmitArgs ea{};
  ea.hits = p.hits.as<uint32_t>();
  ea.counts = p.haveCounts ? p.counts.as<uint32_t>() : nullptr;
  ea.quarterSums = p.haveUnitSums ? p.grpUnitSums.as<uint32_t>() : nullptr;
  ea.next = t.next.as<uint32_t>();
  ea.tileOffsets = p.tileOffsets.as<uint64_t>();
  ea.numRows = p.numRows;
  ea.firstTile = firstTile;
  ea.windowBegin = begin;
  ea.windowEnd = end;
  ea.joinType = p.joinType;
  ea.missValue = p.missValue;
  ea.nullKeyValue = p.nullKeyValue;
  ea.mapping = dMap;
  ea.buildRows = dRows;
  ea.staged = p.sparse ? p.staged.as<uint2>() : nullptr;
  ea.tileDense = p.sparse ? p.tileDense.as<uint8_t>() : nullptr;
  if (!p.filter.empty()) {
    fillFilterArgs(p, p.batch, &ea.f);
  }
  // Build columns the probe staged next to the hits (wide slots) leave with the emit pass; the
  // others are gathered by build row afterwards.
  std::vector<vx355_out_column> restCols;
  std::vector<int32_t> restIds;
  if (p.wideStaged > 0 && !p.haveCounts && numBuildCols > 0) {
    VX_CHECK_ARG(buildCols && buildColIds, "NULL build column arguments");
    bool taken[kWideDeps] = {false, false};
    for (int32_t c = 0; c < numBuildCols; ++c) {
      int slot = -1;
      for (int i = 0; i < p.wideStaged; ++i) {
        if (!taken[i] && buildColIds[c] == t.wideDeps[i] && buildCols[c].mem == VX355_MEM_DEVICE &&
            buildCols[c].type_kind == t.depKinds[t.wideDeps[i]]) {
          slot = i;
        }
      }
      if (slot < 0) {
        restCols.push_back(buildCols[c]);
        restIds.push_back(buildColIds[c]);
        continue;
      }
      taken[slot] = true;
      ea.hitVals[slot] = p.hitVals[slot].as<uint64_t>();
      ea.depVals[slot] = t.depVals[t.wideDeps[slot]].as<uint64_t>();
      ea.outVals[slot] = static_cast<uint64_t*>(buildCols[c].values);
      if (buildCols[c].nulls && n > 0) {
        VX_LAUNCH("k_fill_valid", k_fill_valid, static_cast<int>(ceilDiv(ceilDiv(n, 64), 256)), 256, 0,
                  buildCols[c].nulls, static_cast<int64_t>(n));
      }
    }
    buildCols = restCols.data();
    buildColIds = restIds.data();
    numBuildCols = static_cast<int32_t>(restCols.size());
  }
  VX_LAUNCH("k_emit", k_emit, static_cast<int>(lastTile - firstTile + 1), 256, 0, ea);

  if (numBuildCols > 0) {
    gatherBuildCols(p, t, dRows, n, buildCols, buildColIds, numBuildCols);
  }
  if (host) {
    copyOutAsync(mappingOut, VX355_MEM_HOST, dMap, static_cast<size_t>(n) * 4);
    if (buildRowsOut) {
      copyOutAsync(buildRowsOut, VX355_MEM_HOST, dRows, static_cast<size_t>(n) * 4);
    }
  }
  rt.sync();
  p.cursor = end;
  *nOut = n;
  *finished = p.cursor >= p.totalOut ? 1 : 0;
}


// HashProbe::getBuildSideOutput (HashProbe.cpp:993-1080): listNotProbedRows for
// right / full joins, listProbedRows for right semi filter, ascending row id.
void probeGetBuildSideOutput(vx355_join_probe& p, int32_t maxRows, int32_t* buildRowsOut, int32_t outMem,
                             vx355_out_column* buildCols, const int32_t* buildColIds, int32_t numBuildCols,
                             int32_t* nOut, int32_t* finished) {
  auto& rt = Runtime::get();
  auto& t = *p.table;
  VX_CHECK_ARG(nOut && finished, "NULL argument");
  VX_CHECK_ARG(maxRows > 0, "max_rows must be positive");
  VX_CHECK_ARG(numBuildCols >= 0 && numBuildCols <= kMaxDeps, "bad number of build columns");
  if (!marksProbedRows(p.joinType)) {
    VX_THROW(VX355_EINVAL, "this join type has no build-side output");
  }
  *nOut = 0;
  *finished = 1;
  if (p.buildSideCount < 0) {
    p.buildSideCount = 0;
    if (t.numRows > 0) {
      DevBuf bits, scratch;
      bits.ensure(static_cast<size_t>(ceilDiv(t.numRows, 64)) * 8 + 64);
      VX_LAUNCH("k_probed_bits", k_probed_bits, streamGrid(t.numRows, 256), 256, 0, t.probed.as<uint8_t>(),
                t.numRows,
                p.joinType == VX355_JOIN_RIGHT_SEMI_FILTER ? 1 : (p.joinType == VX355_JOIN_RIGHT_SEMI_PROJECT ? 2 : 0),
                bits.as<uint64_t>());
      p.buildSideRows.ensure(static_cast<size_t>(t.numRows) * 4 + 64);
      compactBits(bits.as<uint64_t>(), nullptr, nullptr, t.numRows, p.buildSideRows.as<int32_t>(), scratch,
                  &p.buildSideCount);
    }
  }
  if (p.buildSideCursor >= p.buildSideCount) {
    return;
  }
  const int32_t n = static_cast<int32_t>(std::min<int64_t>(maxRows, p.buildSideCount - p.buildSideCursor));
  const int32_t* dRows = p.buildSideRows.as<int32_t>() + p.buildSideCursor;
  if (numBuildCols > 0) {
    gatherBuildCols(p, t, dRows, n, buildCols, buildColIds, numBuildCols);
  }
  if (buildRowsOut) {
    copyOutAsync(buildRowsOut, outMem, dRows, static_cast<size_t>(n) * 4);
  }
  rt.sync();
  p.buildSideCursor += n;
  *nOut = n;
  *finished = p.buildSideCursor >= p.buildSideCount ? 1 : 0;
}

// ---- dynamic filters (HashProbe::pushdownDynamicFilters, HashProbe.cpp:408-457) ----
constexpr int64_t kMaxDistinctForValues = 100000;  // VectorHasher::kMaxDistinct (VectorHasher.h:139)

// SplitBlockBloomFilter::makeSaltsVec (common/base/SplitBlockBloomFilter.h:96-121).
__device__ __host__ inline uint32_t bloomSalt(int lanes, int lane) {
  constexpr uint32_t kSalts[8] = {0x2df1424bU, 0x44974d91U, 0x47b6137bU, 0x5c6bfb31U,
                                  0x705495c7U, 0x8824ad5bU, 0x9efc4947U, 0xa2b7289dU};
  return lanes == 8 ? kSalts[lane] : kSalts[2 * lane];
}

// insert(hash): block = ((hash >> 32) * numBlocks) >> 32; per lane the bit
// (salt * uint32(hash)) >> 27 (SplitBlockBloomFilter.h:72-76,91-93,123-126).
__global__ __launch_bounds__(256) void k_bloom_insert(const uint64_t* values, int64_t n, uint32_t* blocks,
                                                       uint64_t numBlocks, int32_t lanes) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t h = twangMix64(values[i]);  // folly::hasher<int64_t> (type/Filter.h:1354-1359)
    uint32_t* block = blocks + (((h >> 32) * numBlocks) >> 32) * lanes;
    const uint32_t low = static_cast<uint32_t>(h);
    for (int l = 0; l < lanes; ++l) {
      const uint32_t bit = 1u << ((bloomSalt(lanes, l) * low) >> 27);
      if (!(__hip_atomic_load(block + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) {
        atomicOr(block + l, bit);
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_bloom_test(ColView col, int64_t numRows, const uint32_t* blocks,
                                                     uint64_t numBlocks, int32_t lanes, const uint64_t* rows,
                                                     uint64_t* rowsOut) {
  const int64_t numWords = (numRows + 63) >> 6;
  const int64_t waveStride = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t w = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; w < numWords;
       w += waveStride) {
    const int64_t row = (w << 6) + lane();
    bool pass = row < numRows && (!rows || bitAt(rows, row)) && !colIsNull(col, row);
    if (pass) {
      const uint64_t h = twangMix64(static_cast<uint64_t>(loadInt64(col, colIndex(col, row))));
      const uint32_t* block = blocks + (((h >> 32) * numBlocks) >> 32) * lanes;
      const uint32_t low = static_cast<uint32_t>(h);
      for (int l = 0; l < lanes; ++l) {
        pass = pass && ((block[l] >> ((bloomSalt(lanes, l) * low) >> 27)) & 1u);
      }
    }
    const uint64_t word = ballot(pass);
    if (lane() == 0) {
      rowsOut[w] = word;
    }
  }
}

// Signed order through an unsigned sort.
__global__ __launch_bounds__(256) void k_flip_sign(const uint64_t* in, uint64_t* out, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    out[i] = in[i] ^ (1ULL << 63);
  }
}

// bit i = sorted[i] starts a run.
__global__ __launch_bounds__(256) void k_run_starts(const uint64_t* sorted, int64_t n, uint64_t* bits) {
  const int64_t numWords = (n + 63) >> 6;
  const int64_t waveStride = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t w = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; w < numWords;
       w += waveStride) {
    const int64_t i = (w << 6) + lane();
    const bool start = i < n && (i == 0 || sorted[i] != sorted[i - 1]);
    const uint64_t word = ballot(start);
    if (lane() == 0) {
      bits[w] = word;
    }
  }
}

__global__ __launch_bounds__(256) void k_gather_flipped(const uint64_t* sorted, const int32_t* idx, int64_t n,
                                                         uint64_t* out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    out[i] = sorted[idx[i]] ^ (1ULL << 63);
  }
}

bool filterableKey(const vx355_join_table& t, int32_t key) {
  const int32_t kind = t.keyKinds.at(key);
  // Right / full builds keep rows with null keys (zero key images): no filter from those.
  return kind >= VX355_TINYINT && kind <= VX355_BIGINT && !t.keepsNullRows;
}

// Distinct non-null build values of one key, ascending (uniqueValues_ of the
// build-side VectorHasher, as a sorted list). Computed once per key.
int64_t distinctKeyValues(vx355_join_table& t, int32_t key) {
  auto& rt = Runtime::get();
  if (t.distinctCount.empty()) {
    t.distinctCount.assign(t.keyKinds.size(), -1);
    t.distinctVals.resize(t.keyKinds.size());
  }
  if (t.distinctCount[key] >= 0) {
    return t.distinctCount[key];
  }
  const int64_t n = t.numRows;
  if (n == 0) {
    t.distinctCount[key] = 0;
    return 0;
  }
  DevBuf flipped, sorted, tmp, bits, idx, scratch;
  flipped.ensure(static_cast<size_t>(n) * 8 + 64);
  sorted.ensure(static_cast<size_t>(n) * 8 + 64);
  VX_LAUNCH("k_flip_sign", k_flip_sign, streamGrid(n, 256), 256, 0, t.keyStore[key].as<uint64_t>(),
            flipped.as<uint64_t>(), n);
  sortKeysU64(flipped.as<uint64_t>(), sorted.as<uint64_t>(), static_cast<size_t>(n), tmp);
  bits.ensure(static_cast<size_t>(ceilDiv(n, 64)) * 8 + 64);
  VX_LAUNCH("k_run_starts", k_run_starts, streamGrid(n, 256), 256, 0, sorted.as<uint64_t>(), n,
            bits.as<uint64_t>());
  idx.ensure(static_cast<size_t>(n) * 4 + 64);
  int64_t total = 0;
  compactBits(bits.as<uint64_t>(), nullptr, nullptr, n, idx.as<int32_t>(), scratch, &total);
  t.distinctVals[key].ensure(static_cast<size_t>(std::max<int64_t>(1, total)) * 8 + 64);
  VX_LAUNCH("k_gather_flipped", k_gather_flipped, streamGrid(total, 256), 256, 0, sorted.as<uint64_t>(),
            idx.as<int32_t>(), total, t.distinctVals[key].as<uint64_t>());
  rt.sync();
  t.distinctCount[key] = total;
  return total;
}

int64_t bloomNumBlocks(int64_t numElements, double falsePositive, int32_t lanes) {
  // SplitBlockBloomFilter::numBlocks (SplitBlockBloomFilter.cpp:27-34), K = lanes.
  const double k = lanes;
  const int64_t numBits =
      static_cast<int64_t>(std::ceil(-k * numElements / std::log(1 - std::pow(falsePositive, 1.0 / k))));
  const int64_t blockBits = 32LL * lanes;
  return (numBits + blockBits - 1) / blockBits;
}

}  // namespace
}  // namespace vx

extern "C" {

int vx355_join_build_create(const vx355_join_build_spec* spec, vx355_join_build** out) {
  VX_API_BEGIN
  Runtime::get().requireInit();
  VX_CHECK_ARG(spec && out, "NULL argument");
  VX_CHECK_ARG(spec->num_keys >= 1 && spec->num_keys <= kMaxKeys, "1..8 join keys supported");
  VX_CHECK_ARG(spec->num_dependents >= 0 && spec->num_dependents <= kMaxDeps,
               "at most 16 build payload columns");
  if (!supportedJoin(spec->join_type) ||
      (spec->null_aware && spec->join_type != VX355_JOIN_ANTI && spec->join_type != VX355_JOIN_LEFT_SEMI_PROJECT)) {
    VX_THROW(VX355_EUNSUPPORTED, "join type " + std::to_string(spec->join_type) +
                                     (spec->null_aware ? " (null aware)" : "") + " not on device");
  }
  if (countingJoin(spec->join_type) && spec->num_dependents != 0) {
    VX_THROW(VX355_EINVAL, "counting joins (INTERSECT ALL / EXCEPT ALL) have no build payload columns");
  }
  if (spec->null_as_value && spec->null_aware) {
    VX_THROW(VX355_EINVAL, "nullAware and nullAsValue are mutually exclusive");  // core/PlanNode.h:3474
  }
  if (spec->drop_duplicates && spec->join_type != VX355_JOIN_LEFT_SEMI_FILTER &&
      spec->join_type != VX355_JOIN_LEFT_SEMI_PROJECT && spec->join_type != VX355_JOIN_ANTI) {
    // HashJoinNode::canDropDuplicates (counting joins keep their counts: they always deduplicate)
    VX_THROW(VX355_EINVAL, "drop_duplicates applies to left semi (filter / project) and anti joins without an extra filter");
  }
  auto h = std::make_unique<vx355_join_build>();
  h->joinType = spec->join_type;
  h->dropDuplicates = spec->drop_duplicates != 0;
  h->nullAsValue = spec->null_as_value != 0;
  h->nullAware = spec->null_aware != 0;
  for (int32_t k = 0; k < spec->num_keys; ++k) {
    const int32_t kind = spec->key_types[k];
    if (kindWidth(kind) < 0) {
      VX_THROW(VX355_EUNSUPPORTED, "join key type " + std::to_string(kind));
    }
    h->keyCols.push_back(spec->key_cols[k]);
    h->keyKinds.push_back(kind);
    h->usedCols.push_back(spec->key_cols[k]);
  }
  for (int32_t d = 0; d < spec->num_dependents; ++d) {
    const int32_t kind = spec->dependent_types[d];
    if (kindWidth(kind) < 0) {
      VX_THROW(VX355_EUNSUPPORTED, "payload type " + std::to_string(kind));
    }
    h->depCols.push_back(spec->dependent_cols[d]);
    h->depKinds.push_back(kind);
    h->usedCols.push_back(spec->dependent_cols[d]);
  }
  h->keyVals.resize(h->keyCols.size());
  h->depVals.resize(h->depCols.size());
  h->depValid.resize(h->depCols.size());
  h->obsMin.assign(h->keyCols.size(), INT64_MAX);
  h->obsMax.assign(h->keyCols.size(), INT64_MIN);
  h->ctx = Runtime::createContext();
  for (int32_t kind : h->keyKinds) {
    h->hasStringCols = h->hasStringCols || isString(kind);
  }
  for (int32_t kind : h->depKinds) {
    h->hasStringCols = h->hasStringCols || isString(kind);
  }
  for (int32_t kind : h->keyKinds) {
    // No value ids for these types (VectorHasher.h:338-357): generic mode whatever the
    // data, also for an empty build side.
    if (kind == VX355_REAL || kind == VX355_DOUBLE || kind == VX355_TIMESTAMP) {
      h->unmappable = true;
    }
  }
  *out = h.release();
  VX_API_END
}

static int joinBuildAddInputNow(vx355_join_build* h, const vx355_batch* batch) {
  VX_API_BEGIN_CTX(VX_CTX_OF(h))
  Runtime::get().requireInit();
  VX_CHECK_ARG(h && batch, "NULL argument");
  VX_CHECK_ARG(!h->finished, "addInput after finish");
  if (h->coalescer.append(batch, h->usedCols)) {
    if (h->coalescer.pendingRows() >= h->coalescer.thresholdRows) {
      h->coalescer.flush([&](const vx355_batch* flat) { buildAddInput(*h, flat); });
    }
  } else {
    h->coalescer.flush([&](const vx355_batch* flat) { buildAddInput(*h, flat); });
    buildAddInput(*h, batch);
  }
  VX_API_END
}

int vx355_join_build_add_input(vx355_join_build* h, const vx355_batch* batch) {
  VX_ASYNC_DRAIN(h)
  return joinBuildAddInputNow(h, batch);
}

// Asynchronous boundary (async.hip): the batch is processed by the handle's worker thread.
int vx355_join_build_add_input_async(vx355_join_build* h, const vx355_batch* batch, int64_t* ticket_out) {
  try {
    if (!h || !batch || (batch->num_cols > 0 && !batch->cols)) {
      vx::setLastError("NULL argument");
      return VX355_EINVAL;
    }
    if (!h->aq) {
      h->aq = vx::asyncCreate();
    }
    if (const int failed = vx::asyncFailed(h->aq)) {
      return failed;  // an earlier batch failed: the handle stays failed (asyncWait)
    }
    const int64_t ticket = vx::asyncSubmitBatch(h->aq, h->ctx->ds, batch, h->usedCols,
                                                [h](const vx355_batch* b) { return joinBuildAddInputNow(h, b); });
    if (ticket_out) {
      *ticket_out = ticket;
    }
    return VX355_OK;
  } catch (const std::exception& e) {
    vx::setLastError(e.what());
    return VX355_EINTERNAL;
  }
}

int vx355_join_build_poll(vx355_join_build* h, int64_t* submitted, int64_t* completed) {
  if (!h) {
    vx::setLastError("NULL argument");
    return VX355_EINVAL;
  }
  if (submitted) {
    *submitted = 0;
  }
  if (completed) {
    *completed = 0;
  }
  if (h->aq) {
    vx::asyncPoll(h->aq, submitted, completed);
  }
  return VX355_OK;
}

int vx355_join_build_wait(vx355_join_build* h) {
  if (!h) {
    vx::setLastError("NULL argument");
    return VX355_EINVAL;
  }
  return h->aq ? vx::asyncWait(h->aq) : VX355_OK;
}

int vx355_join_build_finish(vx355_join_build* h, vx355_join_build* const* others, int32_t num_others,
                            vx355_join_table** out) {
  VX_ASYNC_DRAIN(h)
  for (int32_t i = 0; i < num_others; ++i) {
    if (others && others[i]) {
      VX_ASYNC_DRAIN(others[i])
    }
  }
  VX_API_BEGIN_CTX(VX_CTX_OF(h))
  Runtime::get().requireInit();
  VX_CHECK_ARG(h && out && num_others >= 0, "bad argument");
  h->coalescer.flush([&](const vx355_batch* flat) { buildAddInput(*h, flat); });
  for (int32_t i = 0; i < num_others; ++i) {
    if (others && others[i]) {
      vx355_join_build* o = others[i];
      o->coalescer.flush([&](const vx355_batch* flat) { buildAddInput(*o, flat); });
    }
  }
  *out = buildFinish(*h, others, num_others);
  VX_API_END
}

void* vx355_join_build_stream(vx355_join_build* h) { return h ? static_cast<void*>(h->ctx->stream) : nullptr; }

void vx355_join_build_destroy(vx355_join_build* h) {
  if (!h) {
    return;
  }
  vx::asyncDestroy(h->aq);  // waits for the batches in flight
  h->aq = nullptr;
  Runtime* ctx = h->ctx;
  try {
    vx::ContextScope scope(ctx);
    delete h;
  } catch (...) {
  }
  Runtime::destroyContext(ctx);
}

void vx355_join_table_retain(vx355_join_table* t) {
  if (t) {
    t->refs.fetch_add(1);
  }
}

void vx355_join_table_release(vx355_join_table* t) {
  if (t && t->refs.fetch_sub(1) == 1) {
    // The last reference may go away on any thread (probe destroy, the bridge): every
    // entry point that used the table has drained its stream, so the blocks are idle.
    Runtime* def = nullptr;
    try {
      def = Runtime::defaultContext(t->device);
    } catch (...) {  // after vx355_shutdown: blocks go straight back to the driver
    }
    if (def) {
      vx::ContextScope scope(def);
      delete t;
    } else {
      delete t;
    }
  }
}

int vx355_join_table_get_stats(const vx355_join_table* t, vx355_join_table_stats* out) {
  VX_API_BEGIN
  VX_CHECK_ARG(t && out, "NULL argument");
  out->num_rows = t->numRows;
  out->num_distinct = t->numDistinct;
  out->capacity = static_cast<int64_t>(t->capacity);
  out->hash_mode = t->mode;
  out->has_duplicates = t->hasDuplicates ? 1 : 0;
  VX_API_END
}

int vx355_join_build_get_gpu_stats(const vx355_join_build* h, vx355_gpu_stats* out) {
  return vx::gpuStatsOf(h ? h->ctx : nullptr, out);
}
int vx355_join_probe_get_gpu_stats(const vx355_join_probe* h, vx355_gpu_stats* out) {
  return vx::gpuStatsOf(h ? h->ctx : nullptr, out);
}

int vx355_join_probe_create(vx355_join_table* table, const vx355_join_probe_spec* spec,
                            vx355_join_probe** out) {
  VX_API_BEGIN
  Runtime::get().requireInit();
  VX_CHECK_ARG(table && spec && out, "NULL argument");
  VX_CHECK_ARG(spec->num_keys == static_cast<int32_t>(table->keyKinds.size()),
               "probe and build key counts differ");
  if (!supportedJoin(spec->join_type) ||
      (spec->null_aware && spec->join_type != VX355_JOIN_ANTI && spec->join_type != VX355_JOIN_LEFT_SEMI_PROJECT)) {
    VX_THROW(VX355_EUNSUPPORTED, "join type " + std::to_string(spec->join_type) + " not on device");
  }
  if ((marksProbedRows(spec->join_type) || marksProbedRows(table->joinType)) &&
      spec->join_type != table->joinType) {
    VX_THROW(VX355_EINVAL, "right / full / right semi joins need a table built for that join type");
  }
  if (countingJoin(spec->join_type) != countingJoin(table->joinType) ||
      (countingJoin(spec->join_type) && spec->join_type != table->joinType)) {
    VX_THROW(VX355_EINVAL, "counting joins need a table built for that join type");
  }
  if (Runtime::get().device != table->device) {
    VX_THROW(VX355_EINVAL, "probe created on another GPU than its join table (vx355_set_device)");
  }
  if ((spec->null_as_value != 0) != table->nullAsValue) {
    VX_THROW(VX355_EINVAL, "nullAsValue of the probe and of its join table differ");
  }
  auto p = std::make_unique<vx355_join_probe>();
  p->ctx = Runtime::createContext();
  p->table = table;
  vx355_join_table_retain(table);
  p->joinType = spec->join_type;
  p->nullAware = spec->null_aware != 0;
  if (const char* e = std::getenv("VX355_JOIN_PARTITION")) {
    p->partitionMode = std::atoi(e);  // 0 never, 1 whenever eligible, otherwise adaptive
    if (p->partitionMode != 0 && p->partitionMode != 1) {
      p->partitionMode = -1;
    }
  }
  if (const char* e = std::getenv("VX355_JOIN_PARTITION_FAST")) {
    p->partitionFast = std::atoi(e) != 0;
  }
  if (const char* e = std::getenv("VX355_JOIN_WIDE")) {
    p->wideMode = std::atoi(e);
  }
  if (const char* e = std::getenv("VX355_JOIN_REGROUP")) {
    p->regroupMode = std::atoi(e);
    if (p->regroupMode != 0 && p->regroupMode != 1) {
      p->regroupMode = -1;
    }
  }
  if (const char* e = std::getenv("VX355_JOIN_SLICE_BYTES")) {
    p->sliceBytes = std::max<int64_t>(1 << 16, std::atoll(e));
  }
  if (const char* e = std::getenv("VX355_JOIN_GROUP_UNIT")) {
    p->groupUnit = std::atoi(e) == 8192 ? 8192 : 2048;
  }
  if (const char* e = std::getenv("VX355_JOIN_GROUP_PREFETCH")) {
    p->groupPrefetch = std::atoi(e) != 0;
  }
  if (const char* e = std::getenv("VX355_JOIN_GROUP_WGS")) {
    p->groupWgs = std::max(0, std::atoi(e));
  }
  if (const char* e = std::getenv("VX355_JOIN_WINDOW")) {
    p->window = std::atoi(e) != 0;
  }
  if (const char* e = std::getenv("VX355_JOIN_SPARSE")) {
    p->sparseMode = std::atoi(e);  // 0 never, 1 always, otherwise adaptive
    if (p->sparseMode != 0 && p->sparseMode != 1) {
      p->sparseMode = -1;
    }
  }
  p->keyCols.assign(spec->key_cols, spec->key_cols + spec->num_keys);
  p->usedCols = p->keyCols;
  *out = p.release();
  VX_API_END
}

int vx355_join_probe_set_filter(vx355_join_probe* h, const vx355_join_filter_term* terms, int32_t n_terms) {
  VX_API_BEGIN_CTX(VX_CTX_OF(h))
  VX_CHECK_ARG(h && (terms || n_terms == 0), "NULL argument");
  VX_CHECK_ARG(n_terms >= 0 && n_terms <= kMaxJoinFilterTerms, "0..4 join filter terms");
  VX_CHECK_ARG(!h->hasInput, "set_filter after the first add_input");
  if (n_terms > 0 && countingJoin(h->joinType)) {
    VX_THROW(VX355_EUNSUPPORTED, "counting joins take no extra filter (exec/HashProbe.cpp:1345-1365)");
  }
  if (n_terms > 0 && h->table->droppedDuplicates) {
    // a filter needs every build row of a key (the one that passes may be a duplicate)
    VX_THROW(VX355_EINVAL, "the build side dropped duplicate keys (HashJoinNode::canDropDuplicates requires no filter)");
  }
  h->filter.assign(terms, terms + n_terms);
  h->usedCols = h->keyCols;
  for (const auto& t : h->inputFilter) {
    h->usedCols.push_back(t.col);
  }
  for (const auto& t : h->filter) {
    VX_CHECK_ARG(t.left_side == 0 || t.left_side == 1, "join filter: left_side is 0 (probe) or 1 (build)");
    VX_CHECK_ARG(t.right_kind >= 0 && t.right_kind <= 2, "join filter: bad right_kind");
    VX_CHECK_ARG(t.cmp >= VX355_CMP_EQ && t.cmp <= VX355_CMP_GE, "join filter: bad comparison");
    if (t.left_side == 0) {
      h->usedCols.push_back(t.left_col);
    }
    if (t.right_kind == 1) {
      h->usedCols.push_back(t.right_col);
    }
  }
  VX_API_END
}

int vx355_join_probe_set_input_filter(vx355_join_probe* h, const vx355_filter_term* terms, int32_t n_terms) {
  VX_API_BEGIN_CTX(VX_CTX_OF(h))
  VX_CHECK_ARG(h && (terms || n_terms == 0), "NULL argument");
  VX_CHECK_ARG(n_terms >= 0 && n_terms <= vx::kMaxTerms, "0..4 filter terms");
  VX_CHECK_ARG(!h->hasInput, "set_input_filter after the first add_input");
  if (n_terms > 0 && vx::outputCount(h->joinType, 0) != 0) {
    // LEFT / FULL / LEFT_SEMI_PROJECT / ANTI emit probe rows that found nothing: a row the filter
    // removed must not come out as one of those (the FilterProject stays a separate operator)
    VX_THROW(VX355_EUNSUPPORTED, "input filter fusion: only join kinds whose unmatched probe rows emit nothing");
  }
  if (n_terms > 0 && h->nullAware) {
    VX_THROW(VX355_EUNSUPPORTED, "input filter fusion: not with null-aware joins");
  }
  for (int32_t i = 0; i < n_terms; ++i) {
    VX_CHECK_ARG(terms[i].col >= 0, "input filter: bad column");
    VX_CHECK_ARG(terms[i].cmp >= VX355_CMP_EQ && terms[i].cmp <= VX355_CMP_GE, "input filter: bad comparison");
  }
  h->inputFilter.assign(terms, terms + n_terms);
  for (const auto& t : h->inputFilter) {
    h->usedCols.push_back(t.col);
  }
  VX_API_END
}

int vx355_join_probe_set_output_batch_bytes(vx355_join_probe* h, int64_t bytes) {
  try {
    VX_CHECK_ARG(h && bytes >= 0, "bad argument");
    h->outputBatchBytes = bytes;
  VX_API_CATCH
}

static int joinProbeAddInputNow(vx355_join_probe* h, const vx355_batch* batch) {
  VX_API_BEGIN_CTX(VX_CTX_OF(h))
  Runtime::get().requireInit();
  VX_CHECK_ARG(h && batch, "NULL argument");
  probeAddInput(*h, batch);
  VX_API_END
}

int vx355_join_probe_add_input(vx355_join_probe* h, const vx355_batch* batch) {
  VX_ASYNC_DRAIN(h)
  return joinProbeAddInputNow(h, batch);
}

int vx355_join_probe_add_input_regrouped(vx355_join_probe* h, const vx355_batch* batch, void* const* regrouped_values,
                                         int32_t* regrouped) {
  VX_ASYNC_DRAIN(h)
  VX_API_BEGIN_CTX(VX_CTX_OF(h))
  Runtime::get().requireInit();
  VX_CHECK_ARG(h && batch && regrouped_values && regrouped, "NULL argument");
  probeAddInputRegrouped(*h, batch, regrouped_values, regrouped);
  VX_API_END
}

// Asynchronous boundary for the probe side (exec/Operator.h:285-299: a Driver thread must not sit in
// addInput): the batch's upload, probe kernels and the read-back of the output size run on the
// handle's worker thread. One batch at a time, as for the synchronous form: the next add_input
// (either form) follows the last get_output of this one. Every other entry point waits for it.
int vx355_join_probe_add_input_async(vx355_join_probe* h, const vx355_batch* batch, int64_t* ticket_out) {
  try {
    if (!h || !batch || (batch->num_cols > 0 && !batch->cols)) {
      vx::setLastError("NULL argument");
      return VX355_EINVAL;
    }
    if (!h->aq) {
      h->aq = vx::asyncCreate();
    }
    if (const int failed = vx::asyncFailed(h->aq)) {
      return failed;
    }
    // (never through the parallel ingest: it merges vectors into chunks, and a probe's output rows
    // number the rows of ONE input batch)
    const int64_t ticket = vx::asyncSubmit(
        h->aq, vx::asyncBatchTask(batch, [h](const vx355_batch* b) { return joinProbeAddInputNow(h, b); }));
    if (ticket_out) {
      *ticket_out = ticket;
    }
    return VX355_OK;
  } catch (const std::exception& e) {
    vx::setLastError(e.what());
    return VX355_EINTERNAL;
  }
}

int vx355_join_probe_poll(vx355_join_probe* h, int64_t* submitted, int64_t* completed) {
  if (!h) {
    vx::setLastError("NULL argument");
    return VX355_EINVAL;
  }
  if (submitted) {
    *submitted = 0;
  }
  if (completed) {
    *completed = 0;
  }
  if (h->aq) {
    vx::asyncPoll(h->aq, submitted, completed);
  }
  return VX355_OK;
}

int vx355_join_probe_wait(vx355_join_probe* h) {
  if (!h) {
    vx::setLastError("NULL argument");
    return VX355_EINVAL;
  }
  return h->aq ? vx::asyncWait(h->aq) : VX355_OK;
}

// (the entry points minus their drain: what the queue's worker runs for vx355_join_probe_get_output_async)
static int joinProbeGetOutputNow(vx355_join_probe* h, int32_t max_rows, int32_t* mapping_out,
                                 int32_t* build_rows_out, int32_t out_mem, vx355_out_column* build_cols,
                                 const int32_t* build_col_ids, int32_t num_build_cols, int32_t* n_out,
                                 int32_t* finished) {
  VX_API_BEGIN_CTX(VX_CTX_OF(h))
  Runtime::get().requireInit();
  VX_CHECK_ARG(h, "NULL argument");
  probeGetOutput(*h, max_rows, mapping_out, build_rows_out, out_mem, build_cols, build_col_ids,
                 num_build_cols, n_out, finished);
  VX_API_END
}

int vx355_join_probe_get_output(vx355_join_probe* h, int32_t max_rows, int32_t* mapping_out,
                                int32_t* build_rows_out, int32_t out_mem, vx355_out_column* build_cols,
                                const int32_t* build_col_ids, int32_t num_build_cols, int32_t* n_out,
                                int32_t* finished) {
  VX_ASYNC_DRAIN(h)
  return joinProbeGetOutputNow(h, max_rows, mapping_out, build_rows_out, out_mem, build_cols, build_col_ids,
                               num_build_cols, n_out, finished);
}

static int joinProbeGetBuildSideOutputNow(vx355_join_probe* h, int32_t max_rows, int32_t* build_rows_out,
                                          int32_t out_mem, vx355_out_column* build_cols,
                                          const int32_t* build_col_ids, int32_t num_build_cols, int32_t* n_out,
                                          int32_t* finished) {
  VX_API_BEGIN_CTX(VX_CTX_OF(h))
  Runtime::get().requireInit();
  VX_CHECK_ARG(h, "NULL argument");
  probeGetBuildSideOutput(*h, max_rows, build_rows_out, out_mem, build_cols, build_col_ids, num_build_cols,
                          n_out, finished);
  VX_API_END
}

int vx355_join_probe_get_build_side_output(vx355_join_probe* h, int32_t max_rows, int32_t* build_rows_out,
                                           int32_t out_mem, vx355_out_column* build_cols,
                                           const int32_t* build_col_ids, int32_t num_build_cols, int32_t* n_out,
                                           int32_t* finished) {
  VX_ASYNC_DRAIN(h)
  return joinProbeGetBuildSideOutputNow(h, max_rows, build_rows_out, out_mem, build_cols, build_col_ids,
                                        num_build_cols, n_out, finished);
}

// Queued output page (ABI 7): vx355_join_probe_get_output / _get_build_side_output as a task of the handle's
// worker, behind the batch queued with vx355_join_probe_add_input_async; 'done' fires on the worker thread.
int vx355_join_probe_get_output_async(vx355_join_probe* h, int32_t build_side, int32_t max_rows, int32_t* mapping_out,
                                      int32_t* build_rows_out, int32_t out_mem, const vx355_out_column* build_cols,
                                      const int32_t* build_col_ids, int32_t num_build_cols,
                                      vx355_output_done_fn done, void* done_arg, int64_t* ticket_out) {
  try {
    if (!h || (num_build_cols > 0 && (!build_cols || !build_col_ids))) {
      vx::setLastError("NULL argument");
      return VX355_EINVAL;
    }
    if (!h->aq) {
      h->aq = vx::asyncCreate();
    }
    if (const int failed = vx::asyncFailed(h->aq)) {
      return failed;
    }
    auto page = std::make_shared<vx355_join_probe::QueuedPage>();
    page->cols.assign(build_cols, build_cols + num_build_cols);
    page->ids.assign(build_col_ids, build_col_ids + num_build_cols);
    std::function<void(int)> fire;
    if (done) {
      fire = [page, done, done_arg](int queueStatus) {
        done(done_arg, page->status != VX355_OK ? page->status : queueStatus, page->numRows, page->finished);
      };
    }
    std::lock_guard<std::mutex> lock(h->pagesMutex);
    const int64_t ticket = vx::asyncSubmit(
        h->aq,
        [h, page, build_side, max_rows, mapping_out, build_rows_out, out_mem](std::string* text) {
          const int32_t nc = static_cast<int32_t>(page->cols.size());
          page->status = build_side
              ? joinProbeGetBuildSideOutputNow(h, max_rows, build_rows_out, out_mem, page->cols.data(), page->ids.data(), nc,
                                               &page->numRows, &page->finished)
              : joinProbeGetOutputNow(h, max_rows, mapping_out, build_rows_out, out_mem, page->cols.data(),
                                      page->ids.data(), nc, &page->numRows, &page->finished);
          if (page->status != VX355_OK) {
            *text = vx355_last_error();
            page->errorText = *text;
          }
          page->complete.store(true, std::memory_order_release);
          return page->status;
        },
        std::move(fire));
    h->pages[ticket] = page;
    if (ticket_out) {
      *ticket_out = ticket;
    }
    return VX355_OK;
  } catch (const std::exception& e) {
    vx::setLastError(e.what());
    return VX355_EINTERNAL;
  }
}

int vx355_join_probe_output_result(vx355_join_probe* h, int64_t ticket, int32_t* n_out, int32_t* finished) {
  if (!h || !n_out || !finished) {
    vx::setLastError("NULL argument");
    return VX355_EINVAL;
  }
  std::shared_ptr<vx355_join_probe::QueuedPage> page;
  {
    std::lock_guard<std::mutex> lock(h->pagesMutex);
    auto it = h->pages.find(ticket);
    if (it == h->pages.end()) {
      vx::setLastError("no queued get_output with this ticket (results are handed out once)");
      return VX355_EINVAL;
    }
    // The queue's position FIRST, the page's flag after it (see vx355_agg_output_result).
    int64_t submitted = 0, completed = 0;
    vx::asyncPoll(h->aq, &submitted, &completed);
    if (!it->second->complete.load(std::memory_order_acquire)) {
      if (completed < ticket || vx::asyncFailed(h->aq) == VX355_OK) {
        vx::setLastError("the queued get_output has not completed (vx355_join_probe_poll: completed < ticket)");
        return VX355_EINVAL;
      }
      h->pages.erase(it);  // skipped behind a failed batch: the queue's failure is the answer
      return vx::asyncFailed(h->aq);
    }
    page = it->second;
    h->pages.erase(it);
  }
  if (page->status != VX355_OK) {
    vx::setLastError(page->errorText);
    return page->status;
  }
  *n_out = page->numRows;
  *finished = page->finished;
  return VX355_OK;
}

void vx355_join_probe_destroy(vx355_join_probe* h) {
  if (!h) {
    return;
  }
  vx::asyncDestroy(h->aq);  // waits for a batch in flight
  h->aq = nullptr;
  vx355_join_table* t = h->table;
  Runtime* ctx = h->ctx;
  try {
    vx::ContextScope scope(ctx);
    delete h;
  } catch (...) {
  }
  Runtime::destroyContext(ctx);
  vx355_join_table_release(t);
}

void* vx355_join_probe_stream(vx355_join_probe* h) { return h ? static_cast<void*>(h->ctx->stream) : nullptr; }

int vx355_join_table_key_filter(vx355_join_table* t, int32_t key, vx355_key_filter* out) {
  VX_API_BEGIN_DEV(t ? t->device : -1)
  std::lock_guard<std::mutex> tableLock(t ? t->lazyMutex : vx::gNoTableMutex);
  Runtime::get().requireInit();
  VX_CHECK_ARG(t && out, "NULL argument");
  VX_CHECK_ARG(key >= 0 && key < static_cast<int32_t>(t->keyKinds.size()), "no such key");
  *out = vx355_key_filter{};
  if (!filterableKey(*t, key)) {
    return VX355_OK;  // VectorHasher::getFilter returns nullptr (VectorHasher.cpp:776-778)
  }
  const int64_t distinct = distinctKeyValues(*t, key);
  out->min = t->ranges[key].min;
  out->max = t->ranges[key].max;
  if (distinct <= kMaxDistinctForValues) {
    out->kind = VX355_KEY_FILTER_VALUES;
    out->num_distinct = distinct;
  } else {
    out->kind = VX355_KEY_FILTER_BLOOM;
    out->num_distinct = t->numDistinct;  // HashTable.cpp:966-971 sizes with the table's distinct keys
  }
  VX_API_END
}

int vx355_join_table_key_filter_values(vx355_join_table* t, int32_t key, int64_t* values_out, int64_t capacity,
                                       int32_t mem, int64_t* n_out) {
  VX_API_BEGIN_DEV(t ? t->device : -1)
  std::lock_guard<std::mutex> tableLock(t ? t->lazyMutex : vx::gNoTableMutex);
  Runtime::get().requireInit();
  VX_CHECK_ARG(t && n_out, "NULL argument");
  VX_CHECK_ARG(key >= 0 && key < static_cast<int32_t>(t->keyKinds.size()), "no such key");
  if (!filterableKey(*t, key)) {
    VX_THROW(VX355_EUNSUPPORTED, "no value filter for this key type");
  }
  const int64_t distinct = distinctKeyValues(*t, key);
  *n_out = distinct;
  VX_CHECK_ARG(capacity >= distinct, "values_out too small");
  if (distinct > 0) {
    VX_CHECK_ARG(values_out, "NULL argument");
    copyOut(values_out, mem, t->distinctVals[key].ptr(), static_cast<size_t>(distinct) * 8);
  }
  VX_API_END
}

int64_t vx355_bloom_num_blocks(int64_t num_elements, double false_positive, int32_t lanes) {
  if ((lanes != 4 && lanes != 8) || num_elements < 0 || !(false_positive > 0 && false_positive < 1)) {
    return -1;
  }
  return bloomNumBlocks(num_elements, false_positive, lanes);
}

int vx355_join_table_key_filter_bloom(vx355_join_table* t, int32_t key, int32_t lanes, uint32_t* blocks_out,
                                      int64_t num_blocks, int32_t mem) {
  VX_API_BEGIN_DEV(t ? t->device : -1)
  std::lock_guard<std::mutex> tableLock(t ? t->lazyMutex : vx::gNoTableMutex);
  auto& rt = Runtime::get();
  rt.requireInit();
  VX_CHECK_ARG(t && blocks_out, "NULL argument");
  VX_CHECK_ARG(key >= 0 && key < static_cast<int32_t>(t->keyKinds.size()), "no such key");
  VX_CHECK_ARG(lanes == 4 || lanes == 8, "lanes must be 4 or 8");
  VX_CHECK_ARG(num_blocks > 0, "num_blocks must be positive");
  if (!filterableKey(*t, key)) {
    VX_THROW(VX355_EUNSUPPORTED, "no Bloom filter for this key type");  // supportsBloomFilter, VectorHasher.h:270-283
  }
  const size_t bytes = static_cast<size_t>(num_blocks) * lanes * 4;
  DevBuf scratch;
  uint32_t* blocks = mem == VX355_MEM_HOST ? static_cast<uint32_t*>(scratch.ensure(bytes + 64)) : blocks_out;
  HIP_OK(hipMemsetAsync(blocks, 0, bytes, rt.stream));
  if (t->numRows > 0) {
    VX_LAUNCH("k_bloom_insert", k_bloom_insert, streamGrid(t->numRows, 256), 256, 0,
              t->keyStore[key].as<uint64_t>(), t->numRows, blocks, static_cast<uint64_t>(num_blocks), lanes);
  }
  if (mem == VX355_MEM_HOST) {
    copyOut(blocks_out, VX355_MEM_HOST, blocks, bytes);
  }
  rt.sync();
  VX_API_END
}

int vx355_bloom_test(const uint32_t* blocks, int64_t num_blocks, int32_t lanes, const vx355_column* column,
                     int32_t num_rows, const uint64_t* rows, uint64_t* rows_out, int32_t mem) {
  VX_API_BEGIN
  auto& rt = Runtime::get();
  rt.requireInit();
  VX_CHECK_ARG(blocks && column && rows_out, "NULL argument");
  VX_CHECK_ARG(lanes == 4 || lanes == 8, "lanes must be 4 or 8");
  VX_CHECK_ARG(num_blocks > 0 && num_rows >= 0, "bad sizes");
  VX_CHECK_ARG(column->type_kind >= VX355_TINYINT && column->type_kind <= VX355_BIGINT, "integer columns only");
  if (num_rows == 0) {
    return VX355_OK;
  }
  vx355_batch one{num_rows, 1, column};
  DeviceBatch db;
  db.load(&one, std::vector<int32_t>{0});
  const size_t words = static_cast<size_t>(ceilDiv(num_rows, 64));
  const size_t blockBytes = static_cast<size_t>(num_blocks) * lanes * 4;
  DevBuf dBlocks, dRows, dOut;
  const uint32_t* b = blocks;
  const uint64_t* r = rows;
  uint64_t* o = rows_out;
  if (mem == VX355_MEM_HOST) {
    copyIn(dBlocks.ensure(blockBytes + 64), blocks, VX355_MEM_HOST, blockBytes);
    b = dBlocks.as<uint32_t>();
    if (rows) {
      copyIn(dRows.ensure(words * 8 + 64), rows, VX355_MEM_HOST, words * 8);
      r = dRows.as<uint64_t>();
    }
    o = static_cast<uint64_t*>(dOut.ensure(words * 8 + 64));
  }
  VX_LAUNCH("k_bloom_test", k_bloom_test, streamGrid(num_rows, 256), 256, 0, db.col(0),
            static_cast<int64_t>(num_rows), b, static_cast<uint64_t>(num_blocks), lanes, r, o);
  if (mem == VX355_MEM_HOST) {
    copyOut(rows_out, VX355_MEM_HOST, o, words * 8);
  }
  rt.sync();
  VX_API_END
}

}  // extern "C"
