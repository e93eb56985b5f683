// Device code shared by the ahead-of-time build of agg.hip and the hiprtc
// instantiation of the shape-specialised aggregation kernel (jit.hip): group-row
// constants, accumulator arithmetic, the LDS accumulation helpers and the
// kernel body template. Depends only on device_utils.h / expr_device.h; no
// host-only headers, so hiprtc can compile it.
#pragma once
#include "device_utils.h"
#include "expr_device.h"

namespace vx {

constexpr int kMaxKeys = 8;
constexpr int kMaxAccs = 16;      // accumulators a kernel updates per row
constexpr int kMaxLdsAccs = 28;   // LDS / table words they touch (DOUBLE sums own two)
constexpr uint64_t kEmpty = ~0ULL;
constexpr uint64_t kNoRow = ~0ULL;

enum AccKind : int32_t {
  ACC_SUM_F64 = 0,
  ACC_SUM_I64 = 1,        // sum(BIGINT): LOW word of a 128-bit total, carries go to the next word
  ACC_SUM_I64_WRAP = 2,   // partial counts merged in the final step (no check)
  ACC_COUNT = 3,          // +1 per qualifying row
  ACC_MIN = 4,            // order-preserving u64 image, atomic umin
  ACC_MAX = 5,
  ACC_SUM_I64_HI = 6,     // high word of the 128-bit total behind an ACC_SUM_I64 word: plain wrapping adds
};

// sum(BIGINT) keeps a 128-bit total per group (two words) so that "integer overflow" depends on
// the rows alone, not on the order in which lanes, waves and workgroups add them: the reference
// checks its running sum in input order (vector/AggregationHook.h:126-135 checkedPlus); any
// parallel order sees other partial sums, and a check on those would raise — or not — at random
// on mixed-sign data. Here every add is exact (wrapping low word + carry into the high word,
// both commutative), and the total is checked once, when it is read out: it must fit int64.
__host__ __device__ inline int accWords(int32_t kind) { return (kind == ACC_SUM_F64 || kind == ACC_SUM_I64) ? 2 : 1; }
__host__ __device__ inline int32_t accSecondKind(int32_t kind) { return kind == ACC_SUM_I64 ? ACC_SUM_I64_HI : kind; }

// What adding the signed value v to a low word that held oldLo sends to the high word.
__device__ inline int64_t carrySigned(uint64_t oldLo, int64_t v) {
  const uint64_t nl = oldLo + static_cast<uint64_t>(v);
  return static_cast<int64_t>(nl < oldLo ? 1 : 0) - (v < 0 ? 1 : 0);
}
// Same for an UNSIGNED partial low word (a low word accumulated elsewhere, whose own carries
// already sit in its high word).
__device__ inline int64_t carryUnsigned(uint64_t oldLo, uint64_t lo) { return (oldLo + lo) < oldLo ? 1 : 0; }

enum Mode : int32_t { MODE_HASH = 0, MODE_ARRAY = 1, MODE_NORMALIZED = 2 };

// Counters the kernels bump; mirrored into the pinned mailbox by the host.
struct Counters {
  uint32_t numDeferred;
  uint32_t numNewGroups;
  uint32_t overflow;     // sum(BIGINT) overflowed
  uint32_t unmappable;   // a string key longer than 7 bytes was seen
  uint32_t tableFull;
  uint32_t pairsBroken;  // a radix fold split a partition into slices: its (first row, group) pairs may be stale
  uint32_t firstRowsDistinct;  // distinct keys among the first 2048 rows of the first batch (key statistics)
  uint32_t pad;
  int64_t keyMin[kMaxKeys];
  int64_t keyMax[kMaxKeys];
  uint64_t sumMax[kMaxAccs];  // largest |input| seen per DOUBLE sum (bit pattern), k_sum_stats
};

__device__ inline bool addOverflows(int64_t old, int64_t v) {
  int64_t r;
  return __builtin_add_overflow(old, v, &r);
}

__device__ inline void applyGlobal(uint64_t* word, int32_t kind, uint64_t v, Counters* ctr) {
  switch (kind) {
    case ACC_SUM_F64:
      unsafeAtomicAdd(reinterpret_cast<double*>(word), __longlong_as_double(static_cast<long long>(v)));
      break;
    case ACC_SUM_I64: {
      // v is a signed input value; the high word is the next word of the group row
      const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(word),
                                               static_cast<unsigned long long>(v));
      const int64_t up = carrySigned(old, static_cast<int64_t>(v));
      if (up != 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(word + 1), static_cast<unsigned long long>(up));
      }
      break;
    }
    case ACC_SUM_I64_HI:
    case ACC_SUM_I64_WRAP:
    case ACC_COUNT:
      atomicAdd(reinterpret_cast<unsigned long long*>(word), static_cast<unsigned long long>(v));
      break;
    case ACC_MIN:
      atomicMin(reinterpret_cast<unsigned long long*>(word), static_cast<unsigned long long>(v));
      break;
    default:
      atomicMax(reinterpret_cast<unsigned long long*>(word), static_cast<unsigned long long>(v));
      break;
  }
}

// Adds the 128-bit partial {lo, hi} (lo unsigned) to the total at word / word + 1 in HBM.
__device__ inline void addPartial128Global(uint64_t* word, uint64_t lo, int64_t hi) {
  int64_t up = hi;
  if (lo != 0) {
    const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(word), static_cast<unsigned long long>(lo));
    up += carryUnsigned(old, lo);
  }
  if (up != 0) {
    atomicAdd(reinterpret_cast<unsigned long long*>(word + 1), static_cast<unsigned long long>(up));
  }
}

__host__ __device__ inline uint64_t accIdentity(int32_t kind) {
  return kind == ACC_MIN ? ~0ULL : 0ULL;
}

// Error-free split of v against the grid encoded in m = 1.5 * 2^(G+52):
// hi is v rounded to a multiple of 2^G, lo = v - hi exactly (|v| < 2^(G+51)).
__device__ inline void splitDouble(double v, double m, double* hi, double* lo) {
  if (!(fabs(v) < m * 0.25)) {
    // Too large for the grid (or inf / NaN): plain accumulation for this value.
    *hi = v;
    *lo = 0.0;
    return;
  }
  const double t = v + m;
  *hi = t - m;
  *lo = v - *hi;
}

// ---- low-cardinality kernels: LDS-resident, lane-replicated accumulators ----
// LDS layout: [slotOf int32[capacity] unless direct][slotKey u32[S]][slotFirst u32[S]]
//             [numSlots u32][pad][acc u64[S][numAccs][REP]]
constexpr int32_t kSlotEmpty = -1;
constexpr int32_t kSlotPending = -2;
constexpr int32_t kSlotOverflow = -3;

// Group row for a key in the open-addressing table: linear probing from
// twang_mix64(key) (HashTable.cpp:442-444 mixNormalizedKey); the key word is
// claimed with one CAS and never changes afterwards, so stale reads are safe.
__device__ inline uint64_t* findOrInsert(uint64_t* table, int32_t stride, uint64_t capacity,
                                         uint64_t key, Counters* ctr) {
  const uint64_t mask = capacity - 1;
  uint64_t pos = twangMix64(key) & mask;
  for (uint64_t probes = 0; probes <= mask; ++probes) {
    uint64_t* row = table + pos * stride;
    uint64_t k = __hip_atomic_load(row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == key) {
      return row;
    }
    if (k == kEmpty) {
      unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(row), kEmpty, key);
      if (old == kEmpty || old == key) {
        return row;
      }
    }
    pos = (pos + 1) & mask;
  }
  ctr->tableFull = 1;
  return nullptr;
}

// What both LDS kernels need to know about the accumulators and the table.
struct LdsPlan {
  int32_t S;
  int32_t REP;
  int32_t A;
  int32_t direct;
  uint64_t capacity;
  uint64_t* table;
  int32_t stride;
  int32_t mapWords;   // direct == 2: entries of the hashed key -> slot map (a power of two)
  int32_t tableMode;  // MODE_ARRAY: group row = table + key * stride; MODE_NORMALIZED: findOrInsert(key)
  // Hashed slot map over an open-addressing table (direct == 2): a row whose workgroup has run out
  // of LDS slots is DEFERRED (replayed by the host) instead of inserted straight into the table.
  // New groups per launch are then bounded by workgroups x slots, so the table is sized for that -
  // not for "every row of the chunk is a new group" (a 12 GB table to initialise and to scan for
  // Q1's 600 M rows and 196 groups).
  int32_t deferOverflow;
  uint64_t rowBase;
  Counters* counters;
  // Direct-index tables with many live keys (BASELINE config 1: 1000 groups): instead of one HBM
  // atomic per (key, word) and WORKGROUP, every workgroup stores its reduced words side by side in
  // scratch[workgroup][word][key] (plain coalesced stores) and k_lds_reduce folds the workgroups'
  // copies into the table: the flush costs bytes, not atomics (3 M atomics -> 32 MB). nullptr = atomics.
  uint64_t* scratch;
  int32_t kind[kMaxLdsAccs];
  int32_t off[kMaxLdsAccs];
};

struct LdsState {
  int32_t* slotOf;
  uint32_t* slotKey;
  uint32_t* slotFirst;
  uint32_t* numSlots;
  uint64_t* slotKey64;   // direct == 2: the slots' full normalized keys
  uint64_t* acc;
};

// The group row of a normalized key in the operator's table.
__device__ inline uint64_t* ldsGroupRow(const LdsPlan& p, uint64_t key) {
  if (p.tableMode == MODE_NORMALIZED) {
    return findOrInsert(p.table, p.stride, p.capacity, key, p.counters);
  }
  return p.table + key * p.stride;
}

__device__ inline LdsState ldsInit(const LdsPlan& p, unsigned char* raw) {
  LdsState st;
  // key -> slot map: none (direct == 1: slot = key), one entry per possible key (0), or an open-
  // addressing table over the keys that occur (2: key ranges too wide for an entry per key)
  const int mapWords = p.direct == 1 ? 0 : (p.direct == 2 ? p.mapWords : static_cast<int>(p.capacity));
  st.slotOf = reinterpret_cast<int32_t*>(raw);
  st.slotKey = reinterpret_cast<uint32_t*>(raw) + mapWords;
  st.slotFirst = st.slotKey + p.S;
  st.numSlots = st.slotFirst + p.S;
  st.slotKey64 = reinterpret_cast<uint64_t*>(
      raw + ((static_cast<size_t>(mapWords + 2 * p.S + 2) * 4 + 15) & ~static_cast<size_t>(15)));
  st.acc = st.slotKey64 + (p.direct == 2 ? p.S : 0);
  for (int i = threadIdx.x; i < mapWords; i += blockDim.x) {
    st.slotOf[i] = kSlotEmpty;
  }
  for (int i = threadIdx.x; i < p.S; i += blockDim.x) {
    st.slotFirst[i] = 0xffffffffu;
    st.slotKey[i] = static_cast<uint32_t>(i);
  }
  if (threadIdx.x == 0) {
    *st.numSlots = p.direct == 1 ? static_cast<uint32_t>(p.S) : 0;
  }
  for (int i = threadIdx.x; i < p.S * p.A * p.REP; i += blockDim.x) {
    st.acc[i] = accIdentity(p.kind[(i / p.REP) % p.A]);
  }
  blockSync();
  return st;
}

// LDS slot of a key (>= 0) or kSlotOverflow when the workgroup's slots are used up.
__device__ inline int32_t ldsSlot(const LdsPlan& p, const LdsState& st, uint64_t key) {
  if (p.direct == 1) {
    return static_cast<int32_t>(key);
  }
  if (p.direct == 2) {
    // Few groups spread over a wide key range (keys of two or three characters, sparse integer
    // codes): the map is a hash table of slot numbers, the slot's key (slotKey) confirms a hit.
    // Same claim protocol as below; an entry holding another key sends the lane to the next one.
    const uint32_t mask = static_cast<uint32_t>(p.mapWords) - 1;
    uint32_t pos = static_cast<uint32_t>((key * 0x9E3779B97F4A7C15ULL) >> 40) & mask;
    uint32_t probes = 0;
    int32_t slot = kSlotPending;
    while (slot == kSlotPending) {
      int32_t* entry = st.slotOf + pos;
      int32_t s = __hip_atomic_load(entry, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (s == kSlotEmpty) {
        if (atomicCAS(entry, kSlotEmpty, kSlotPending) == kSlotEmpty) {
          uint32_t t = atomicAdd(st.numSlots, 1u);
          if (t < static_cast<uint32_t>(p.S)) {
            st.slotKey64[t] = key;
            s = static_cast<int32_t>(t);
          } else {
            s = kSlotOverflow;
          }
          __hip_atomic_store(entry, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          s = kSlotPending;
        }
      } else if (s >= 0 && st.slotKey64[s] != key) {
        pos = (pos + 1) & mask;
        s = ++probes > mask ? kSlotOverflow : kSlotPending;
      }
      slot = s;
    }
    return slot;
  }
  // Claim protocol without waiting on an exit edge: the winner of the CAS
  // allocates and publishes the slot INSIDE the loop body, every lane
  // re-evaluates at the latch. (A `break` after the publish would let the
  // compiler sink the publish behind the loop and spin the other lanes of the
  // same wave forever.)
  int32_t* entry = st.slotOf + key;
  int32_t slot = kSlotPending;
  while (slot == kSlotPending) {
    int32_t s = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (s == kSlotEmpty) {
      if (atomicCAS(entry, kSlotEmpty, kSlotPending) == kSlotEmpty) {
        uint32_t t = atomicAdd(st.numSlots, 1u);
        if (t < static_cast<uint32_t>(p.S)) {
          st.slotKey[t] = static_cast<uint32_t>(key);
          s = static_cast<int32_t>(t);
        } else {
          s = kSlotOverflow;
        }
        __hip_atomic_store(entry, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        s = kSlotPending;
      }
    }
    slot = s;
  }
  return slot;
}

__device__ inline void ldsTouchFirst(const LdsState& st, int32_t slot, uint32_t row) {
  if (st.slotFirst[slot] > row) {
    atomicMin(&st.slotFirst[slot], row);
  }
}

// The replicas of LDS word j of a slot, reduced (what one workgroup contributes to the group row).
// ACC_SUM_I64 low words are unsigned partials: how often their sum wraps does not depend on the
// order of the adds, so the thread of the HIGH word counts the carries of word j - 1 itself.
__device__ inline uint64_t ldsReduceReplicas(const LdsPlan& p, const LdsState& st, int slot, int j) {
  const int A = p.A, REP = p.REP;
  const int32_t kind = p.kind[j];
  const uint64_t* q = st.acc + (static_cast<size_t>(slot) * A + j) * REP;
  uint64_t v = q[0];
  if (kind == ACC_SUM_F64) {
    double s = __longlong_as_double(static_cast<long long>(v));
    for (int r = 1; r < REP; ++r) {
      s += __longlong_as_double(static_cast<long long>(q[r]));
    }
    return static_cast<uint64_t>(__double_as_longlong(s));
  }
  if (kind == ACC_MIN) {
    for (int r = 1; r < REP; ++r) {
      v = q[r] < v ? q[r] : v;
    }
    return v;
  }
  if (kind == ACC_MAX) {
    for (int r = 1; r < REP; ++r) {
      v = q[r] > v ? q[r] : v;
    }
    return v;
  }
  for (int r = 1; r < REP; ++r) {
    v += q[r];
  }
  if (kind == ACC_SUM_I64_HI) {
    const uint64_t* lowWords = q - REP;  // word j - 1 of the same slot
    uint64_t low = 0;
    for (int r = 0; r < REP; ++r) {
      v += carryUnsigned(low, lowWords[r]);
      low += lowWords[r];
    }
  }
  return v;
}

// Scratch flush (LdsPlan::scratch): element t = word * capacity + key of this workgroup's copy; keys
// the workgroup never saw store the word's identity (and kNoRow as first row).
__device__ inline void ldsFlushScratch(const LdsPlan& p, const LdsState& st) {
  blockSync();
  const int A = p.A;
  const int R = static_cast<int>(p.capacity);
  uint64_t* out = p.scratch + static_cast<size_t>(blockIdx.x) * R * (A + 1);
  for (int t = threadIdx.x; t < R * (A + 1); t += blockDim.x) {
    const int j = t / R;
    const int key = t - j * R;
    const int32_t slot = p.direct == 1 ? key : st.slotOf[key];
    const uint32_t first = slot >= 0 ? st.slotFirst[slot] : 0xffffffffu;
    uint64_t v;
    if (first == 0xffffffffu) {
      v = j == A ? kNoRow : accIdentity(p.kind[j]);
    } else if (j == A) {
      v = p.rowBase + static_cast<uint64_t>(first);
    } else {
      v = ldsReduceReplicas(p, st, slot, j);
    }
    out[t] = v;
  }
}

// Flush: one (slot, accumulator) pair per thread; replicas reduced in LDS.
__device__ inline void ldsFlush(const LdsPlan& p, const LdsState& st) {
  if (p.scratch != nullptr) {
    ldsFlushScratch(p, st);
    return;
  }
  blockSync();
  const int S = p.S, A = p.A, REP = p.REP;
  uint32_t live = *st.numSlots;
  if (live > static_cast<uint32_t>(S)) {
    live = S;
  }
  for (int t = threadIdx.x; t < static_cast<int>(live) * (A + 1); t += blockDim.x) {
    const int slot = t / (A + 1);
    const int j = t % (A + 1);
    const uint32_t first = st.slotFirst[slot];
    if (first == 0xffffffffu) {
      continue;  // direct layout: key never seen by this workgroup
    }
    uint64_t* g = ldsGroupRow(p, p.direct == 2 ? st.slotKey64[slot] : static_cast<uint64_t>(st.slotKey[slot]));
    if (g == nullptr) {
      continue;  // table full: flagged in the counters
    }
    if (j == A) {
      // 'first' is the smallest ORIGINAL row of the chunk seen for this key
      // (replays go through the row list), so it decides the group order.
      uint64_t firstRow = p.rowBase + static_cast<uint64_t>(first);
      unsigned long long old = atomicMin(reinterpret_cast<unsigned long long*>(g + 1), firstRow);
      if (old == kNoRow) {
        atomicAdd(&p.counters->numNewGroups, 1u);
      }
      continue;
    }
    const int32_t kind = p.kind[j];
    const uint64_t* q = st.acc + (static_cast<size_t>(slot) * A + j) * REP;
    uint64_t v = q[0];
    if (kind == ACC_SUM_F64) {
      double s = __longlong_as_double(static_cast<long long>(v));
      for (int r = 1; r < REP; ++r) {
        s += __longlong_as_double(static_cast<long long>(q[r]));
      }
      applyGlobal(g + p.off[j], ACC_SUM_F64, static_cast<uint64_t>(__double_as_longlong(s)), p.counters);
    } else if (kind == ACC_MIN) {
      for (int r = 1; r < REP; ++r) {
        v = q[r] < v ? q[r] : v;
      }
      applyGlobal(g + p.off[j], ACC_MIN, v, p.counters);
    } else if (kind == ACC_MAX) {
      for (int r = 1; r < REP; ++r) {
        v = q[r] > v ? q[r] : v;
      }
      applyGlobal(g + p.off[j], ACC_MAX, v, p.counters);
    } else if (kind == ACC_SUM_I64) {
      // low words of the replicas (their carries sit in the replicas' high words, flushed by
      // the thread of word j + 1): unsigned adds, the carries join the high word in HBM
      uint64_t s = v;
      int64_t up = 0;
      for (int r = 1; r < REP; ++r) {
        up += carryUnsigned(s, q[r]);
        s += q[r];
      }
      addPartial128Global(g + p.off[j], s, up);
    } else {
      uint64_t s = v;
      for (int r = 1; r < REP; ++r) {
        s += q[r];
      }
      applyGlobal(g + p.off[j], kind == ACC_COUNT ? ACC_SUM_I64_WRAP : kind, s, p.counters);
    }
  }
}

// hiStride: distance (in words) from an ACC_SUM_I64 word to its high word in this LDS layout.
__device__ inline void applyLds(uint64_t* word, int32_t kind, uint64_t v, Counters* ctr, int hiStride = 1) {
  switch (kind) {
    case ACC_SUM_F64:
      unsafeAtomicAdd(reinterpret_cast<double*>(word), __longlong_as_double(static_cast<long long>(v)));
      break;
    case ACC_SUM_I64: {
      const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(word),
                                               static_cast<unsigned long long>(v));
      const int64_t up = carrySigned(old, static_cast<int64_t>(v));
      if (up != 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(word + hiStride), static_cast<unsigned long long>(up));
      }
      break;
    }
    case ACC_SUM_I64_HI:
    case ACC_SUM_I64_WRAP:
    case ACC_COUNT:
      atomicAdd(reinterpret_cast<unsigned long long*>(word), static_cast<unsigned long long>(v));
      break;
    case ACC_MIN:
      atomicMin(reinterpret_cast<unsigned long long*>(word), static_cast<unsigned long long>(v));
      break;
    default:
      atomicMax(reinterpret_cast<unsigned long long*>(word), static_cast<unsigned long long>(v));
      break;
  }
}




// ---- shape-specialised LDS kernel ------------------------------------------------
// The generic LDS kernel interprets the plan per row (column encodings, types,
// masks, expression tables): ~500 VALU instructions per 64 rows, which caps it
// near 2 TB/s. For flat, null-free inputs the plan SHAPE is lifted into template
// parameters instead, so every register index and every branch on the plan is
// resolved at compile time and what is left per row is the arithmetic itself:
//   keys   K0,K1  : FK_VIEW (short string, first 8 bytes of the StringView),
//                   FK_I32, FK_I64, or -1 (absent)
//   terms  T0,T1  : filter column kinds FK_I32 / FK_I64 / FK_F64 or -1
//   NL            : distinct DOUBLE columns loaded per row (each loaded once)
//   NA, ACC_LO/HI : per accumulator 16 bits {numFactors, load index of factor
//                   0..2 (15 = constant factor)}; numFactors 0 = count(*)
// Comparison operators, constants, scales, offsets, ranges stay runtime values.
// Shapes are instantiated ahead of time below (VX_FAST_SHAPES); a plan whose
// shape is not in the table runs on the generic kernel.
constexpr int kFastKeys = 4;
constexpr int kFastTerms = 2;
constexpr int kFastAccs = 12;
constexpr int kFastFactors = 3;
constexpr int kFastLoads = 8;

enum FastKind : int32_t { FK_NONE = -1, FK_VIEW = 0, FK_I32 = 1, FK_I64 = 2, FK_F64 = 3 };

struct FastTerm {
  const void* ptr;
  int32_t cmp;
  int32_t pad;
  int64_t i64;
  double f64;
};
struct FastArgs {
  const void* keyPtr[kFastKeys];
  KeyRange range[kFastKeys];
  const void* loadPtr[kFastLoads];   // element type per the shape's LK mask
  FastTerm term[kFastTerms];
  double scale[kFastAccs][kFastFactors];
  double offset[kFastAccs][kFastFactors];
  double splitM[kFastAccs];  // grid of the hi/lo split of sum j (0 = accumulate into hi only)
  int64_t numRows;
  int32_t* deferred;
  uint32_t deferCap;
  uint32_t pad;
  // Dictionary-wrapped inputs (what FilterProject hands downstream,
  // exec/OperatorUtils.cpp:393-422): the columns flagged in the shape's IND mask
  // are read at indices[row] instead of row; one shared index vector.
  const int32_t* indices;
  // Null bitmaps (1 = valid, one bit per top-level row) of the columns flagged in the shape's
  // NUL mask: a null key is id 0 (or drops the row: ignoreNullKeys), a null filter input fails
  // the filter, a null aggregate input skips that accumulator for the row - what
  // SimpleNumericAggregate::updateGroups does in the same pass
  // (functions/lib/aggregates/SimpleNumericAggregate.h:94-160). One bit per row and column:
  // free in bandwidth next to the 8-byte values.
  const uint64_t* keyNulls[kFastKeys];
  const uint64_t* termNulls[kFastTerms];
  const uint64_t* loadNulls[kFastLoads];
  // AggregationMasks of the accumulators flagged in the shape's MSK mask: flat BOOLEAN columns
  // (bit-packed values + optional null bitmap); a false or null mask skips the accumulator for the row.
  const uint64_t* maskBits[kFastAccs];
  const uint64_t* maskNulls[kFastAccs];
  int32_t ignoreNullKeys;
  int32_t pad2;
  LdsPlan plan;
};

// Element types of the loaded operand columns (FastShape::LK, two bits per load): every operand
// reaches the arithmetic as a double, as loadDouble() hands it to the interpreting kernel.
enum FastLoadKind : int32_t { FL_F64 = 0, FL_F32 = 1, FL_I64 = 2, FL_I32 = 3 };
// FastShape::OPS. FO_DEFAULT: count (no factors) or DOUBLE sum of the product of the factors; the
// others take ONE unscaled operand: checked BIGINT sum of an integer column (128-bit total), min /
// max on the order-preserving image of a floating or of an integer column.
enum FastOp : int32_t { FO_DEFAULT = 0, FO_SUM_I64 = 1, FO_MIN_F = 2, FO_MAX_F = 3, FO_MIN_I = 4, FO_MAX_I = 5 };

constexpr uint64_t accDesc(int numFactors, int l0 = 15, int l1 = 15, int l2 = 15) {
  return static_cast<uint64_t>(numFactors) | (static_cast<uint64_t>(l0) << 4) |
      (static_cast<uint64_t>(l1) << 8) | (static_cast<uint64_t>(l2) << 12);
}
constexpr uint64_t packAccs(uint64_t a0 = 0, uint64_t a1 = 0, uint64_t a2 = 0, uint64_t a3 = 0) {
  return a0 | (a1 << 16) | (a2 << 32) | (a3 << 48);
}

// Bit of key k in the IND / NUL masks: keys 0 and 1 sit at bits 0 and 1 (terms at 2, 3, loads at
// 4..11), keys 2 and 3 behind the loads at bits 12 and 13.
__host__ __device__ constexpr int fastKeyBit(int k) { return k < 2 ? k : 10 + k; }
// KX: kinds of the third and fourth key, four bits each (0xf = absent) - BASELINE's "4-key" Q1.
constexpr uint32_t kFastNoExtraKeys = 0xffu;
constexpr uint32_t packExtraKeys(int k2, int k3) {
  return (static_cast<uint32_t>(k2) & 15u) | ((static_cast<uint32_t>(k3) & 15u) << 4);
}

template <int UNROLL, int K0, int K1, int T0, int T1, int NL, int NA, uint64_t ACC_LO, uint64_t ACC_HI,
          uint32_t IND = 0, uint32_t NUL = 0, uint64_t ACC_EX = 0, uint32_t LK = 0, uint32_t MSK = 0, uint64_t OPS = 0,
          uint32_t KX = kFastNoExtraKeys>
struct FastShape {
  static constexpr int loadKind(int j) { return static_cast<int>((LK >> (2 * j)) & 3); }
  static constexpr bool masked(int j) { return (MSK >> j) & 1; }
  // OPS, four bits per accumulator: what is done with its (single, unscaled) operand - FastOp.
  static constexpr int op(int j) { return static_cast<int>((OPS >> (4 * j)) & 15); }
  static constexpr bool intLoad(int j) { return loadKind(j) == 2 || loadKind(j) == 3; }
  static constexpr bool anyIntLoad = (LK & 0xaaaau) != 0;
  // IND bit fastKeyBit(k): key k, bit 2 + t: filter term t, bit 4 + j: loaded column j is dictionary wrapped.
  static constexpr bool indirect(int bit) { return (IND >> bit) & 1; }
  static constexpr bool anyIndirect = IND != 0;
  // NUL, same bit numbering: the column carries a null bitmap.
  static constexpr bool nullable(int bit) { return (NUL >> bit) & 1; }
  static constexpr int unroll = UNROLL;
  static constexpr int extraKind(int nibble) { return nibble == 15 ? static_cast<int>(FK_NONE) : nibble; }
  static constexpr int keyKind(int k) {
    return k == 0 ? K0 : (k == 1 ? K1 : extraKind(static_cast<int>((KX >> (4 * (k - 2))) & 15u)));
  }
  static constexpr int termKind(int t) { return t == 0 ? T0 : T1; }
  static constexpr int numLoads = NL;
  static constexpr int numAccs = NA;
  static constexpr uint64_t desc(int j) {
    return ((j < 4 ? ACC_LO >> (16 * j) : (j < 8 ? ACC_HI >> (16 * (j - 4)) : ACC_EX >> (16 * (j - 8))))) & 0xffff;
  }
  static constexpr int numFactors(int j) { return static_cast<int>(desc(j) & 15); }
  // LDS / table word of accumulator j: every DOUBLE sum (hi, lo) and every BIGINT sum (low, carry)
  // before it owns two words, counts / min / max one.
  static constexpr int ldsIndex(int j) {
    int idx = 0;
    for (int q = 0; q < j; ++q) {
      idx += (numFactors(q) == 0 || op(q) >= 2) ? 1 : 2;
    }
    return idx;
  }
  static constexpr int numLdsAccs = ldsIndex(NA);
  static constexpr int load(int j, int f) { return static_cast<int>((desc(j) >> (4 + 4 * f)) & 15); }
};

// Compile-time loop: f(std::integral_constant<int, 0>{}) ... f(<N-1>).
template <int V>
struct IntC {
  static constexpr int value = V;
};
template <int... Is>
struct IntSeq {};
template <int N, int... Is>
struct MakeIntSeq : MakeIntSeq<N - 1, N - 1, Is...> {};
template <int... Is>
struct MakeIntSeq<0, Is...> {
  using type = IntSeq<Is...>;
};
template <int... Is, typename F>
__device__ inline void staticForImpl(IntSeq<Is...>, F&& f) {
  (f(IntC<Is>{}), ...);
}
template <int N, typename F>
__device__ inline void staticFor(F&& f) {
  staticForImpl(typename MakeIntSeq<N>::type{}, f);
}

// Columns the kernel reads exactly once stream past the caches (nontemporal loads): k_agg_fast on
// TPC-H Q1 SF100 6.95 -> 6.64 ms on the same box. -DVX355_FAST_CACHED_LOADS restores plain loads.
#ifdef VX355_FAST_CACHED_LOADS
#define VX355_FAST_LOAD(p) (*(p))
#else
#define VX355_FAST_LOAD(p) __builtin_nontemporal_load(p)
#endif

template <int KIND>
__device__ inline uint64_t fastLoadRaw(const void* ptr, int64_t row) {
  if constexpr (KIND == FK_VIEW) {
    return VX355_FAST_LOAD(static_cast<const uint64_t*>(ptr) + row * 2);
  } else if constexpr (KIND == FK_I32) {
    return VX355_FAST_LOAD(static_cast<const uint32_t*>(ptr) + row);  // sign-extended at use
  } else {
    return VX355_FAST_LOAD(static_cast<const uint64_t*>(ptr) + row);
  }
}

// int64 image of a key (VectorHasher::toInt64 / stringAsNumber); INT64_MIN for
// strings the fast path does not decode (> 3 bytes: the replay reads the view).
template <int KIND>
__device__ inline int64_t fastKeyValue(uint64_t raw) {
  if constexpr (KIND == FK_VIEW) {
    const uint32_t size = static_cast<uint32_t>(raw);
    const uint32_t bytes = static_cast<uint32_t>(raw >> 32);
    const uint32_t shift = (size & 3u) * 8;
    const int64_t v = static_cast<int64_t>((bytes & ((1u << shift) - 1)) + (size ? (1u << shift) : 0u));
    return size > 3 ? INT64_MIN : v;
  } else if constexpr (KIND == FK_I32) {
    return static_cast<int64_t>(static_cast<int32_t>(static_cast<uint32_t>(raw)));
  } else {
    return static_cast<int64_t>(raw);
  }
}

template <typename S>
__device__ inline void aggFastBody(const FastArgs& a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
  constexpr int UNROLL = S::unroll;
  constexpr int NA = S::numAccs;     // accumulators of the plan
  constexpr int A = S::numLdsAccs;   // words they own (DOUBLE sums: hi + lo)
  const LdsPlan& p = a.plan;
  const LdsState st = ldsInit(p, ldsRaw);
  const int REP = p.REP;
  const int rep = lane() & (REP - 1);
  const int64_t tile = static_cast<int64_t>(blockDim.x) * UNROLL;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * tile;
  const int64_t rounds = (a.numRows + stride - 1) / stride;
  int64_t base = static_cast<int64_t>(blockIdx.x) * tile + threadIdx.x;
  for (int64_t r = 0; r < rounds; ++r, base += stride) {
    uint64_t kraw[UNROLL][kFastKeys];
    uint64_t traw[UNROLL][kFastTerms];
    double x[UNROLL][S::numLoads > 0 ? S::numLoads : 1];
    int64_t xi[S::anyIntLoad ? UNROLL : 1][S::numLoads > 0 ? S::numLoads : 1];  // integer columns, unconverted
    // Phase 1: every load of this iteration, column by column, UNROLL rows
    // back to back. Rows past the end are clamped (and ignored in phase 2) so
    // that no load sits under a per-lane predicate.
    int64_t rowc[UNROLL];
    int64_t rowi[UNROLL];  // base-vector row of the dictionary-wrapped columns
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t row = base + static_cast<int64_t>(u) * blockDim.x;
      rowc[u] = row < a.numRows ? row : a.numRows - 1;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      rowi[u] = rowc[u];
      if constexpr (S::anyIndirect) {
        rowi[u] = a.indices[rowc[u]];
      }
    }
    staticFor<kFastKeys>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (S::keyKind(k) != FK_NONE) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
          kraw[u][k] = fastLoadRaw<S::keyKind(k)>(a.keyPtr[k], S::indirect(fastKeyBit(k)) ? rowi[u] : rowc[u]);
        }
      }
    });
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if constexpr (S::termKind(0) != FK_NONE) {
        traw[u][0] = fastLoadRaw<S::termKind(0)>(a.term[0].ptr, S::indirect(2) ? rowi[u] : rowc[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if constexpr (S::termKind(1) != FK_NONE) {
        traw[u][1] = fastLoadRaw<S::termKind(1)>(a.term[1].ptr, S::indirect(3) ? rowi[u] : rowc[u]);
      }
    }
    staticFor<S::numLoads>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t at = S::indirect(4 + j) ? rowi[u] : rowc[u];
        if constexpr (S::loadKind(j) == FL_F32) {
          x[u][j] = static_cast<double>(VX355_FAST_LOAD(static_cast<const float*>(a.loadPtr[j]) + at));
        } else if constexpr (S::loadKind(j) == FL_I64) {
          xi[u][j] = VX355_FAST_LOAD(static_cast<const int64_t*>(a.loadPtr[j]) + at);
          x[u][j] = static_cast<double>(xi[u][j]);
        } else if constexpr (S::loadKind(j) == FL_I32) {
          xi[u][j] = VX355_FAST_LOAD(static_cast<const int32_t*>(a.loadPtr[j]) + at);
          x[u][j] = static_cast<double>(xi[u][j]);
        } else {
          x[u][j] = VX355_FAST_LOAD(static_cast<const double*>(a.loadPtr[j]) + at);
        }
      }
    });
    // null flags of the nullable columns: bit (4 + j) of nul[u] set = load j is null, bit
    // fastKeyBit(k) = key k, bit 2 + t = filter input t (the 64 lanes of a wave share each bitmap word)
    uint32_t nul[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      nul[u] = 0;
      staticFor<kFastKeys>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (S::keyKind(k) != FK_NONE && S::nullable(fastKeyBit(k))) {
          nul[u] |= ((a.keyNulls[k][rowc[u] >> 6] >> (rowc[u] & 63)) & 1) ? 0u : (1u << fastKeyBit(k));
        }
      });
      staticFor<kFastTerms>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (S::termKind(t) != FK_NONE && S::nullable(2 + t)) {
          nul[u] |= ((a.termNulls[t][rowc[u] >> 6] >> (rowc[u] & 63)) & 1) ? 0u : (1u << (2 + t));
        }
      });
      staticFor<S::numLoads>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (S::nullable(4 + j)) {
          nul[u] |= ((a.loadNulls[j][rowc[u] >> 6] >> (rowc[u] & 63)) & 1) ? 0u : (1u << (4 + j));
        }
      });
    }
    // Phase 2: filter, key, LDS updates.
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t row = base + static_cast<int64_t>(u) * blockDim.x;
      bool live = row < a.numRows && (nul[u] & 0xcu) == 0;  // a null filter input fails the filter
      staticFor<kFastTerms>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (S::termKind(t) == FK_F64) {
          live = live && compareValues<double>(a.term[t].cmp,
                                               __longlong_as_double(static_cast<long long>(traw[u][t])),
                                               a.term[t].f64);
        } else if constexpr (S::termKind(t) == FK_I32) {
          // the host only picks this shape when the constant fits int32
          live = live && compareValues<int32_t>(a.term[t].cmp,
                                                static_cast<int32_t>(static_cast<uint32_t>(traw[u][t])),
                                                static_cast<int32_t>(a.term[t].i64));
        } else if constexpr (S::termKind(t) == FK_I64) {
          live = live && compareValues<int64_t>(a.term[t].cmp, static_cast<int64_t>(traw[u][t]),
                                                a.term[t].i64);
        }
      });
      uint64_t key = 0;
      bool defer = false;
      staticFor<kFastKeys>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (S::keyKind(k) != FK_NONE) {
          const int64_t v = fastKeyValue<S::keyKind(k)>(kraw[u][k]);
          if (S::nullable(fastKeyBit(k)) && ((nul[u] >> fastKeyBit(k)) & 1)) {
            live = live && !a.ignoreNullKeys;  // else: a null key is value id 0
          } else if (v < a.range[k].min || v > a.range[k].max) {
            if (live) {
              defer = true;
              if (v != INT64_MIN) {
                atomicMin(reinterpret_cast<long long*>(&p.counters->keyMin[k]), static_cast<long long>(v));
                atomicMax(reinterpret_cast<long long*>(&p.counters->keyMax[k]), static_cast<long long>(v));
              }
            }
          } else {
            key += a.range[k].multiplier *
                (static_cast<uint64_t>(v) - static_cast<uint64_t>(a.range[k].min) + 1);
          }
        }
      });
      if (live && !defer) {
        const int32_t slot = ldsSlot(p, st, key);
        double vals[NA > 0 ? NA : 1];
        uint64_t opv[NA > 0 ? NA : 1];  // operand word of the accumulators with an op (FastOp)
        bool have[NA > 0 ? NA : 1];  // false: some input of accumulator j is null in this row
        staticFor<NA>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          double acc = 0;
          have[j] = true;
          if constexpr (S::masked(j)) {
            const uint64_t bit = 1ULL << (rowc[u] & 63);
            have[j] = (a.maskBits[j][rowc[u] >> 6] & bit) != 0 &&
                (a.maskNulls[j] == nullptr || (a.maskNulls[j][rowc[u] >> 6] & bit) != 0);
          }
          staticFor<S::numFactors(j)>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            double v = a.offset[j][f];
            if constexpr (S::load(j, f) != 15) {
              v = a.scale[j][f] * x[u][S::load(j, f)] + a.offset[j][f];
              if constexpr (S::nullable(4 + S::load(j, f))) {
                have[j] = have[j] && !((nul[u] >> (4 + S::load(j, f))) & 1);
              }
            }
            acc = f == 0 ? v : acc * v;
          });
          // count(x) / count(projection): numFactors 0 with up to three load indexes = rows where
          // none of those columns is null
          if constexpr (S::numFactors(j) == 0) {
            staticFor<kFastFactors>([&](auto gc) {
              constexpr int g = decltype(gc)::value;
              if constexpr (S::load(j, g) != 15) {
                if constexpr (S::nullable(4 + S::load(j, g))) {
                  have[j] = have[j] && !((nul[u] >> (4 + S::load(j, g))) & 1);
                }
              }
            });
          }
          vals[j] = acc;
          opv[j] = 0;
          if constexpr (S::op(j) == FO_SUM_I64) {
            opv[j] = static_cast<uint64_t>(xi[S::anyIntLoad ? u : 0][S::load(j, 0)]);
          } else if constexpr (S::op(j) == FO_MIN_I || S::op(j) == FO_MAX_I) {
            opv[j] = int64ToOrdered(xi[S::anyIntLoad ? u : 0][S::load(j, 0)]);
          } else if constexpr (S::op(j) == FO_MIN_F || S::op(j) == FO_MAX_F) {
            opv[j] = doubleToOrdered(x[u][S::load(j, 0)]);
          }
        });
        if (slot >= 0) {
          ldsTouchFirst(st, slot, static_cast<uint32_t>(row));
          uint64_t* dst = st.acc + (static_cast<size_t>(slot) * A) * REP + rep;
          staticFor<NA>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int w = S::ldsIndex(j);
            if (!have[j]) {
              return;
            }
            if constexpr (S::numFactors(j) == 0) {
              atomicAdd(reinterpret_cast<unsigned long long*>(dst + w * REP), 1ULL);
            } else if constexpr (S::op(j) == FO_SUM_I64) {
              applyLds(dst + w * REP, ACC_SUM_I64, opv[j], p.counters, REP);
            } else if constexpr (S::op(j) == FO_MIN_F || S::op(j) == FO_MIN_I) {
              atomicMin(reinterpret_cast<unsigned long long*>(dst + w * REP), static_cast<unsigned long long>(opv[j]));
            } else if constexpr (S::op(j) == FO_MAX_F || S::op(j) == FO_MAX_I) {
              atomicMax(reinterpret_cast<unsigned long long*>(dst + w * REP), static_cast<unsigned long long>(opv[j]));
            } else {
              if (a.splitM[j] != 0.0) {
                double hi, lo;
                splitDouble(vals[j], a.splitM[j], &hi, &lo);
                unsafeAtomicAdd(reinterpret_cast<double*>(dst + w * REP), hi);
                unsafeAtomicAdd(reinterpret_cast<double*>(dst + (w + 1) * REP), lo);
              } else {
                unsafeAtomicAdd(reinterpret_cast<double*>(dst + w * REP), vals[j]);
              }
            }
          });
        } else if (p.deferOverflow) {
          defer = true;  // workgroup out of LDS slots, table sized for the slots only: the host replays the row
        } else {
          // Workgroup out of LDS slots: straight to the group row in HBM.
          uint64_t* g = ldsGroupRow(p, key);
          if (g != nullptr) {  // (nullptr: table full, flagged in the counters)
          const uint64_t myRow = p.rowBase + static_cast<uint64_t>(row);
          unsigned long long old = atomicMin(reinterpret_cast<unsigned long long*>(g + 1), myRow);
          if (old == kNoRow) {
            atomicAdd(&p.counters->numNewGroups, 1u);
          }
          staticFor<NA>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int w = S::ldsIndex(j);
            if (!have[j]) {
              return;
            }
            if constexpr (S::numFactors(j) == 0) {
              applyGlobal(g + p.off[w], ACC_SUM_I64_WRAP, 1, p.counters);
            } else if constexpr (S::op(j) == FO_SUM_I64) {
              applyGlobal(g + p.off[w], ACC_SUM_I64, opv[j], p.counters);
            } else if constexpr (S::op(j) == FO_MIN_F || S::op(j) == FO_MIN_I) {
              applyGlobal(g + p.off[w], ACC_MIN, opv[j], p.counters);
            } else if constexpr (S::op(j) == FO_MAX_F || S::op(j) == FO_MAX_I) {
              applyGlobal(g + p.off[w], ACC_MAX, opv[j], p.counters);
            } else {
              double hi = vals[j], lo = 0;
              if (a.splitM[j] != 0.0) {
                splitDouble(vals[j], a.splitM[j], &hi, &lo);
              }
              applyGlobal(g + p.off[w], ACC_SUM_F64, static_cast<uint64_t>(__double_as_longlong(hi)),
                          p.counters);
              if (a.splitM[j] != 0.0) {
                applyGlobal(g + p.off[w + 1], ACC_SUM_F64, static_cast<uint64_t>(__double_as_longlong(lo)),
                            p.counters);
              }
            }
          });
          }
        }
      }
      // Rows the fast path cannot place go to the deferred list (one atomic per wave).
      const uint64_t m = ballot(defer);
      if (m != 0) {
        const int leader = __ffsll(static_cast<long long>(m)) - 1;
        uint32_t at = 0;
        if (lane() == leader) {
          at = atomicAdd(&p.counters->numDeferred, static_cast<uint32_t>(popc64(m)));
        }
        at = __shfl(at, leader, kWave);
        if (defer && at + lanePrefix(m) < a.deferCap) {
          a.deferred[at + lanePrefix(m)] = static_cast<int32_t>(row);
        }
      }
    }
  }
  ldsFlush(p, st);
}


}  // namespace vx
