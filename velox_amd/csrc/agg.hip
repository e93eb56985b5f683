// HashAggregation on the MI355X (replaces exec/HashAggregation.cpp:191-424,
// exec/GroupingSet.cpp:190-365,810-884 and the accumulator loops of
// functions/lib/aggregates/*).
//
// Design (not a translation of the reference's row-wise RowContainer):
//  * Group state lives in ONE array of fixed-stride "group rows" in HBM:
//      [normalized key u64][first input row u64][8-byte accumulators ...]
//    indexed directly by the normalized key (array mode: what the reference
//    calls kArray, HashTable.cpp:560-607, but sized for 288 GB of HBM instead
//    of a CPU cache) or by an open-addressing slot found with one CAS on the
//    key word (what the reference calls kNormalizedKey, :523-557). All words of
//    a group share one HBM sector, so a row update is one random sector.
//  * Keys are normalized exactly like VectorHasher in range mode
//    (VectorHasher.cpp:196-224): id = value - min + 1, null = 0, combined with
//    per-key multipliers. Ranges are chosen on the host with the reference's
//    50 % reserve (VectorHasher.cpp:786-835). A value outside the range does
//    not abort the batch: the row is appended to a deferred list, the host
//    widens the range, re-keys the table on device and replays only those rows
//    (the reference instead re-decides the mode and redoes the whole batch,
//    HashTable.cpp:2633-2677).
//  * Low-cardinality group-bys (TPC-H Q1, BASELINE config 1) never touch HBM
//    atomics per row: each workgroup keeps lane-replicated accumulators in LDS
//    (ds_add_f64 / ds_add_u64 / ds_min_u64 / ds_max_u64) behind a small
//    key -> LDS-slot map and flushes once at the end.
//  * First-seen group order (RowContainer order, GroupingSet.cpp:828-839) is
//    reproduced by recording the minimum global input row per group and
//    sorting groups by it at output time.
#include "common.h"
#include "agg_device.h"

#include <dlfcn.h>
#include <glob.h>
#include <sys/stat.h>
#include <unistd.h>
#include <hip/hiprtc.h>
#include "expr_device.h"

#include <algorithm>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <type_traits>
#include <chrono>
#include <cmath>
#include <future>
#include <thread>
#include <utility>
#include <cstdio>
#include <cstdlib>
#include <limits>

namespace vx {

void sortPairsU64U32(uint64_t* keys, uint32_t* vals, uint64_t* keysTmp, uint32_t* valsTmp,
                     size_t n, DevBuf& tmp, bool* resultInTmp, int endBit = 64);
void makeTermArgs(const DeviceBatch& db, const vx355_filter_term* terms, int32_t n, TermArg* out);
void makeProjectionArgs(const DeviceBatch& db, const vx355_projection* proj, int32_t n,
                        ProjectionArg* out);

namespace {




struct KeyArg {
  ColView col;
  KeyRange range;
};

struct AccArg {
  ColView in;
  ColView mask;
  int32_t kind;
  int32_t hasIn;
  int32_t hasMask;
  int32_t inIsInt;  // value travels as int64 (else double)
  int32_t off;      // word offset inside the group row
  int32_t inProj;   // >= 0: the input is projection inProj of the fused FilterProject
  // DOUBLE sums are kept as two words: 'hi' receives the value rounded to a
  // fixed grid (v + splitM) - splitM — sums of grid multiples are exact, hence
  // order independent — and 'lo' (the next word) the exact remainder v - hi.
  // splitM = 0: plain accumulation into hi.
  double splitM;
  int32_t ldsIdx;   // position of 'hi' among the LDS accumulators of the launch
  int32_t phys;     // index of the accumulator in vx355_agg::phys (host bookkeeping)
};


struct AggArgs {
  KeyArg keys[kMaxKeys];
  AccArg accs[kMaxAccs];
  int32_t numKeys;
  int32_t numAccs;
  int32_t ignoreNullKeys;
  int32_t mode;  // MODE_ARRAY: group row index = key; else open addressing
  int64_t numRows;
  const int32_t* rowList;  // replay of deferred rows, or nullptr
  uint64_t rowBase;
  uint64_t* table;
  int32_t stride;  // words per group row
  uint64_t capacity;    // group rows in table (array: range product; hash: power of two)
  int32_t* deferred;
  uint32_t deferCap;            // entries the deferred list can hold
  uint32_t pad2;
  const KeyRange* rescanOld;    // non-null: process only rows outside THESE ranges
  Counters* counters;
  // Fused FilterProject (vx355_agg_set_fused_input): rows failing the filter
  // are skipped, projections feed accumulators directly from the scan columns.
  TermArg terms[kMaxTerms];
  ProjectionArg proj[kMaxProjections];
  int32_t numTerms;
  int32_t numProj;
};

__device__ inline bool predicate(const AccArg& a, int64_t row) {
  if (a.hasMask) {
    // AggregationMasks: a false or null mask excludes the row.
    if (colIsNull(a.mask, row)) {
      return false;
    }
    if (!loadInt64(a.mask, colIndex(a.mask, row))) {
      return false;
    }
  }
  if (a.hasIn && colIsNull(a.in, row)) {
    return false;
  }
  return true;
}

// The 8-byte operand of the accumulator update for this row.
__device__ inline uint64_t operand(const AccArg& a, int64_t row) {
  switch (a.kind) {
    case ACC_COUNT:
      return 1;
    case ACC_SUM_F64: {
      double d = loadDouble(a.in, colIndex(a.in, row));
      return static_cast<uint64_t>(__double_as_longlong(d));
    }
    case ACC_SUM_I64:
    case ACC_SUM_I64_WRAP:
      return static_cast<uint64_t>(loadInt64(a.in, colIndex(a.in, row)));
    default:  // MIN / MAX
      if (a.inIsInt) {
        return int64ToOrdered(loadInt64(a.in, colIndex(a.in, row)));
      }
      return doubleToOrdered(loadDouble(a.in, colIndex(a.in, row)));
  }
}

// Predicate and operand of one accumulator for one row in a single call;
// projections are evaluated from the scan columns.
__device__ inline bool accInput(const AggArgs& a, const AccArg& acc, int64_t row, uint64_t* out) {
  if (acc.inProj < 0) {
    if (!predicate(acc, row)) {
      return false;
    }
    *out = operand(acc, row);
    return true;
  }
  AccArg maskOnly = acc;
  maskOnly.hasIn = 0;
  if (acc.hasMask && !predicate(maskOnly, row)) {
    return false;
  }
  bool valid = true;
  const double d = evalProjection(a.proj[acc.inProj], row, &valid);
  if (!valid) {
    return false;
  }
  switch (acc.kind) {
    case ACC_COUNT:
      *out = 1;
      break;
    case ACC_SUM_F64:
      *out = static_cast<uint64_t>(__double_as_longlong(d));
      break;
    default:
      *out = doubleToOrdered(d);
      break;
  }
  return true;
}



// Normalized key of one input row (VectorHasher::computeValueIds semantics).
// Returns 0 = key ready, 1 = row dropped (null key with ignoreNullKeys),
// 2 = some value lies outside the current ranges (statistics updated).
__device__ inline int normalizedKey(const AggArgs& a, int64_t row, uint64_t* keyOut) {
  uint64_t key = 0;
  bool outside = false;
  bool outsideOld = false;
  for (int k = 0; k < a.numKeys; ++k) {
    const KeyArg& ka = a.keys[k];
    if (colIsNull(ka.col, row)) {
      if (a.ignoreNullKeys) {
        return 1;
      }
      continue;  // null contributes id 0
    }
    int64_t value;
    bool mappable;
    uint64_t id = valueIdAt(ka.col, colIndex(ka.col, row), ka.range, &value, &mappable);
    if (!mappable) {
      a.counters->unmappable = 1;
      outside = true;
      outsideOld = true;
      continue;
    }
    if (a.rescanOld && ka.col.kind != VX355_BOOLEAN &&
        (value < a.rescanOld[k].min || value > a.rescanOld[k].max)) {
      outsideOld = true;
    }
    if (id == 0) {
      outside = true;
      atomicMin(reinterpret_cast<long long*>(&a.counters->keyMin[k]), static_cast<long long>(value));
      atomicMax(reinterpret_cast<long long*>(&a.counters->keyMax[k]), static_cast<long long>(value));
      continue;
    }
    key += ka.range.multiplier * id;
  }
  if (a.rescanOld && !outsideOld) {
    return 1;  // already aggregated by the launch that overflowed its deferred list
  }
  *keyOut = key;
  return outside ? 2 : 0;
}

// Appends this lane's row to the deferred list with one atomic per wave.
__device__ inline void deferRow(const AggArgs& a, bool defer, int32_t row) {
  uint64_t m = ballot(defer);
  if (m == 0) {
    return;
  }
  uint32_t base = 0;
  if (lane() == __ffsll(static_cast<long long>(m)) - 1) {
    base = atomicAdd(&a.counters->numDeferred, static_cast<uint32_t>(popc64(m)));
  }
  base = __shfl(base, __ffsll(static_cast<long long>(m)) - 1, kWave);
  // Past the capacity only the count keeps growing: the host then rescans the
  // chunk instead of replaying the list.
  if (defer && base + lanePrefix(m) < a.deferCap) {
    a.deferred[base + lanePrefix(m)] = row;
  }
}

__device__ inline uint64_t* groupRow(const AggArgs& a, uint64_t key) {
  if (a.mode == MODE_ARRAY) {
    return a.table + key * a.stride;
  }
  return findOrInsert(a.table, a.stride, a.capacity, key, a.counters);
}

// Direct HBM update of one input row (high-cardinality path and LDS overflow).
__device__ inline void updateGlobal(const AggArgs& a, int64_t row, uint64_t key, uint32_t* newGroups) {
  uint64_t* g = groupRow(a, key);
  if (!g) {
    return;
  }
  const uint64_t myRow = a.rowBase + static_cast<uint64_t>(row);
  if (__hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > myRow) {
    unsigned long long old = atomicMin(reinterpret_cast<unsigned long long*>(g + 1), myRow);
    if (old == kNoRow) {
      ++*newGroups;
    }
  }
  for (int i = 0; i < a.numAccs; ++i) {
    const AccArg& acc = a.accs[i];
    uint64_t v;
    if (accInput(a, acc, row, &v)) {
      if (acc.kind == ACC_SUM_F64 && acc.splitM != 0.0) {
        double hi, lo;
        splitDouble(__longlong_as_double(static_cast<long long>(v)), acc.splitM, &hi, &lo);
        applyGlobal(g + acc.off, ACC_SUM_F64, static_cast<uint64_t>(__double_as_longlong(hi)), a.counters);
        applyGlobal(g + acc.off + 1, ACC_SUM_F64, static_cast<uint64_t>(__double_as_longlong(lo)),
                    a.counters);
      } else {
        applyGlobal(g + acc.off, acc.kind, v, a.counters);
      }
    }
  }
}

// Wave-wide reduction of one accumulator operand over the lanes in 'members'
// (others contribute the identity); every lane returns the total.
__device__ inline uint64_t waveCombine(int32_t kind, uint64_t v, bool member, Counters* ctr) {
  uint64_t x = member ? v : accIdentity(kind);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint64_t o = shfl64(x, lane() ^ off);
    switch (kind) {
      case ACC_SUM_F64:
        x = static_cast<uint64_t>(__double_as_longlong(__longlong_as_double(static_cast<long long>(x)) +
                                                        __longlong_as_double(static_cast<long long>(o))));
        break;
      case ACC_SUM_I64_WRAP:
      case ACC_COUNT:
        x += o;
        break;
      case ACC_MIN:
        x = o < x ? o : x;
        break;
      default:
        x = o > x ? o : x;
        break;
    }
  }
  return x;
}

// Wave-wide 128-bit sum of the members' signed values: {lo, hi} on every lane.
__device__ inline void waveCombine128(uint64_t v, bool member, uint64_t* loOut, int64_t* hiOut) {
  uint64_t lo = member ? v : 0;
  int64_t hi = (member && static_cast<int64_t>(v) < 0) ? -1 : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint64_t olo = shfl64(lo, lane() ^ off);
    const int64_t ohi = static_cast<int64_t>(shfl64(static_cast<uint64_t>(hi), lane() ^ off));
    hi += ohi + carryUnsigned(lo, olo);
    lo += olo;
  }
  *loOut = lo;
  *hiOut = hi;
}

// updateGlobal for a whole wave (called by all 64 lanes, 'active' says which
// rows count). One HBM address retires ~88 M atomics/s, so rows of a hot key
// must not each bring their own atomic: the lanes that share the first active
// lane's key — when there are at least kHotLanes of them — are reduced in
// registers and the leader applies one update per accumulator. Everything
// else takes the per-row path.
constexpr int kHotLanes = 8;

__device__ inline void updateGlobalWave(const AggArgs& a, int64_t row, uint64_t key, bool active,
                                        uint32_t* newGroups) {
  const uint64_t act = ballot(active);
  if (act == 0) {
    return;
  }
  // Cheap screen first: a key frequent enough to matter makes neighbouring lanes
  // agree; random keys never do, and then the rounds below are skipped.
  const bool pairEqual = active && key == shfl64(key, lane() ^ 1);
  if (popc64(ballot(pairEqual)) < kHotLanes) {
    if (active) {
      updateGlobal(a, row, key, newGroups);
    }
    return;
  }
  // Up to four candidate leaders: with half of the rows on one key the chance
  // that none of them carries it is 1/16.
  uint64_t candidates = act;
  uint64_t done = 0;
  for (int round = 0; round < 4 && candidates != 0; ++round) {
    const int leader = __ffsll(static_cast<long long>(candidates)) - 1;
    const uint64_t leaderKey = shfl64(key, leader);
    const bool member = active && !((done >> lane()) & 1) && key == leaderKey;
    const uint64_t same = ballot(member);
    if (popc64(same) < kHotLanes) {
      candidates &= ~(1ULL << leader);  // stays on the per-row path
      continue;
    }
    uint64_t* g = nullptr;
    if (lane() == leader) {
      g = groupRow(a, key);
    }
    g = reinterpret_cast<uint64_t*>(shfl64(reinterpret_cast<uint64_t>(g), leader));
    if (g != nullptr) {  // null: table full, flagged by findOrInsert
      const uint64_t firstRow =
          waveCombine(ACC_MIN, a.rowBase + static_cast<uint64_t>(row), member, a.counters);
      if (lane() == leader && __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > firstRow) {
        const unsigned long long old = atomicMin(reinterpret_cast<unsigned long long*>(g + 1), firstRow);
        if (old == kNoRow) {
          ++*newGroups;
        }
      }
      for (int i = 0; i < a.numAccs; ++i) {
        const AccArg& acc = a.accs[i];
        uint64_t v = 0;
        const bool has = member && accInput(a, acc, row, &v);
        const uint64_t any = ballot(has);
        if (acc.kind == ACC_SUM_F64 && acc.splitM != 0.0) {
          // Grid multiples add up exactly in any order; the remainders are tiny.
          double hi = 0.0, lo = 0.0;
          if (has) {
            splitDouble(__longlong_as_double(static_cast<long long>(v)), acc.splitM, &hi, &lo);
          }
          const uint64_t totalHi =
              waveCombine(ACC_SUM_F64, static_cast<uint64_t>(__double_as_longlong(hi)), has, a.counters);
          const uint64_t totalLo =
              waveCombine(ACC_SUM_F64, static_cast<uint64_t>(__double_as_longlong(lo)), has, a.counters);
          if (any != 0 && lane() == leader) {
            applyGlobal(g + acc.off, ACC_SUM_F64, totalHi, a.counters);
            applyGlobal(g + acc.off + 1, ACC_SUM_F64, totalLo, a.counters);
          }
          continue;
        }
        if (acc.kind == ACC_SUM_I64) {
          uint64_t lo;
          int64_t hi;
          waveCombine128(v, has, &lo, &hi);
          if (any != 0 && lane() == leader) {
            addPartial128Global(g + acc.off, lo, hi);
          }
          continue;
        }
        const uint64_t total = waveCombine(acc.kind, v, has, a.counters);
        if (any != 0 && lane() == leader) {
          applyGlobal(g + acc.off, acc.kind, total, a.counters);
        }
      }
    }
    done |= same;
    candidates &= ~same;
  }
  if (active && !((done >> lane()) & 1)) {
    updateGlobal(a, row, key, newGroups);
  }
}

__device__ inline void addNewGroups(Counters* ctr, uint32_t mine) {
  uint32_t total = mine;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    total += __shfl_xor(total, off, kWave);
  }
  if (lane() == 0 && total) {
    atomicAdd(&ctr->numNewGroups, total);
  }
}

// ---- high-cardinality kernel: one lane per row, HBM atomics ----------------
__global__ __launch_bounds__(256) void k_agg_global(AggArgs a) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  uint32_t newGroups = 0;
  const int64_t rounds = (a.numRows + stride - 1) / stride;
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (int64_t r = 0; r < rounds; ++r, i += stride) {
    bool defer = false;
    bool active = false;
    int32_t row = 0;
    uint64_t key = 0;
    if (i < a.numRows) {
      row = a.rowList ? a.rowList[i] : static_cast<int32_t>(i);
      int st = (a.numTerms && !evalFilter(a.terms, a.numTerms, row)) ? 1 : normalizedKey(a, row, &key);
      active = st == 0;
      defer = st == 2;
    }
    updateGlobalWave(a, row, key, active, &newGroups);
    deferRow(a, defer, row);
  }
  addNewGroups(a.counters, newGroups);
}

struct LdsArgs {
  AggArgs a;
  LdsPlan plan;
};
static_assert(sizeof(LdsArgs) <= 4096, "kernel arguments are limited to 4 KB");

// Generic LDS kernel: any column encoding / type / mask the ABI admits.
__global__ __launch_bounds__(1024) void k_agg_lds(LdsArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
  const AggArgs& a = args.a;
  const LdsPlan& p = args.plan;
  const LdsState st = ldsInit(p, ldsRaw);
  const int A = p.A, REP = p.REP;
  const int rep = lane() & (REP - 1);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t rounds = (a.numRows + stride - 1) / stride;
  uint32_t newGroups = 0;
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (int64_t r = 0; r < rounds; ++r, i += stride) {
    bool defer = false;
    int32_t row = 0;
    if (i < a.numRows) {
      row = a.rowList ? a.rowList[i] : static_cast<int32_t>(i);
      uint64_t key;
      int state = (a.numTerms && !evalFilter(a.terms, a.numTerms, row)) ? 1 : normalizedKey(a, row, &key);
      if (state == 2) {
        defer = true;
      } else if (state == 0) {
        const int32_t slot = ldsSlot(p, st, key);
        if (slot >= 0) {
          ldsTouchFirst(st, slot, static_cast<uint32_t>(row));
          uint64_t* base = st.acc + (static_cast<size_t>(slot) * A) * REP + rep;
          for (int j = 0; j < a.numAccs; ++j) {
            const AccArg& ac = a.accs[j];
            uint64_t v;
            if (accInput(a, ac, row, &v)) {
              if (ac.kind == ACC_SUM_F64 && ac.splitM != 0.0) {
                double hi, lo;
                splitDouble(__longlong_as_double(static_cast<long long>(v)), ac.splitM, &hi, &lo);
                applyLds(base + ac.ldsIdx * REP, ACC_SUM_F64,
                         static_cast<uint64_t>(__double_as_longlong(hi)), a.counters);
                applyLds(base + (ac.ldsIdx + 1) * REP, ACC_SUM_F64,
                         static_cast<uint64_t>(__double_as_longlong(lo)), a.counters);
              } else {
                applyLds(base + ac.ldsIdx * REP, ac.kind, v, a.counters, REP);
              }
            }
          }
        } else if (p.deferOverflow) {
          defer = true;
        } else {
          updateGlobal(a, row, key, &newGroups);
        }
      }
    }
    deferRow(a, defer, row);
  }
  ldsFlush(p, st);
  addNewGroups(a.counters, newGroups);
}

template <typename S>
__global__ __launch_bounds__(512, 4) void k_agg_fast(FastArgs a) {
  aggFastBody<S>(a);
}

// Counters on their way to the host without a launch of their own: the LAST workgroup of a launch
// that finishes (a ticket per workgroup) copies the live counters into the context's pinned mailbox
// and puts the pristine values back - what k_read_reset_counters does as a launch of its own
// (BASELINE config 1: two of the eight dispatches of a 130-us step). live == nullptr: not wanted.
struct CounterMail {
  uint64_t* live;
  const uint64_t* pristine;
  uint64_t* mailbox;
  uint32_t* ticket;
};

// Called by the first wave of EVERY workgroup once the workgroup's counter updates are behind it
// (block-synchronised by the caller where other waves took part).
__device__ inline void publishCountersFromLastBlock(const CounterMail& m, uint32_t workgroups) {
  if (m.live == nullptr || threadIdx.x >= 64) {
    return;
  }
  uint32_t t = 0;
  if (lane() == 0) {
    __threadfence();
    t = atomicAdd(m.ticket, 1u);
  }
  t = __shfl(t, 0, kWave);
  if (t != workgroups - 1) {
    return;
  }
  __threadfence();
  constexpr int kWords = static_cast<int>(sizeof(Counters) / 8);
  for (int i = lane(); i < kWords; i += kWave) {
    const uint64_t v = __hip_atomic_load(m.live + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(m.mailbox + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(m.live + i, m.pristine[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (lane() == 0) {
    __hip_atomic_store(m.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Second half of the scratch flush (ldsFlushScratch, agg_device.h): folds the workgroups' copies
// scratch[copy][word][key] into the direct-index table. A block owns 64 consecutive (word, key)
// elements; its 16 waves read them from every 16th copy (512-byte coalesced loads, ~numCopies / 16
// independent loads per lane), the partial results meet in LDS and wave 0 applies the total to the
// group row: one update per (key, word) of the table instead of one per (key, word) and workgroup.
// storeAll: the table has never been written (vx355_agg::tableVirgin) and one block column covers
// every copy: the totals are STORED, identities included - the launch that would have initialised
// the table (k_init_table) is not needed. Otherwise gridDim.y block columns share the copies and
// meet in the table with atomics.
__global__ __launch_bounds__(1024) void k_lds_reduce(LdsPlan p, int32_t numCopies, int32_t storeAll, CounterMail mail) {
  __shared__ uint64_t part[16][64];
  __shared__ uint64_t partLow[16][64];
  const int R = static_cast<int>(p.capacity);
  const int A = p.A;
  const int64_t E = static_cast<int64_t>(R) * (A + 1);
  const int l = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 64 + l;
  const bool in = e < E;
  const int j = in ? static_cast<int>(e / R) : 0;
  const int key = in ? static_cast<int>(e - static_cast<int64_t>(j) * R) : 0;
  const int32_t kind = (!in || j == A) ? static_cast<int32_t>(ACC_MIN) : p.kind[j];  // first rows: a minimum
  uint64_t v = accIdentity(kind);
  uint64_t low = 0;  // ACC_SUM_I64_HI: running sum of the LOW word's copies, whose carries belong here
  if (in) {
    const uint64_t* src = p.scratch + e;
    const int step = 16 * static_cast<int>(gridDim.y);
    // kReduceBatch copies per lane in flight: the loop is a chain of HBM round trips (32 copies per
    // wave for 512 workgroups: eight trips with the four loads a plain unrolled loop kept in flight
    // = 17 - 19 us for BASELINE config 1's 16 MB; round 6)
    constexpr int kReduceBatch = 16;
    const uint64_t identity = accIdentity(kind);
    for (int c = static_cast<int>(blockIdx.y) * 16 + w; c < numCopies; c += step * kReduceBatch) {
      uint64_t q[kReduceBatch], ql[kReduceBatch];
#pragma unroll
      for (int i = 0; i < kReduceBatch; ++i) {
        const int cc = c + i * step;
        q[i] = cc < numCopies ? src[static_cast<int64_t>(cc) * E] : identity;
      }
      if (kind == ACC_SUM_I64_HI) {
#pragma unroll
        for (int i = 0; i < kReduceBatch; ++i) {
          const int cc = c + i * step;
          ql[i] = cc < numCopies ? src[static_cast<int64_t>(cc) * E - R] : 0;
        }
      }
#pragma unroll
      for (int i = 0; i < kReduceBatch; ++i) {
        if (kind == ACC_SUM_F64) {
          if (c + i * step < numCopies) {  // (not "+ 0.0": a sum of nothing but -0.0 keeps its sign)
            v = static_cast<uint64_t>(__double_as_longlong(__longlong_as_double(static_cast<long long>(v)) +
                                                           __longlong_as_double(static_cast<long long>(q[i]))));
          }
        } else if (kind == ACC_MIN) {
          v = q[i] < v ? q[i] : v;
        } else if (kind == ACC_MAX) {
          v = q[i] > v ? q[i] : v;
        } else {
          v += q[i];
          if (kind == ACC_SUM_I64_HI) {
            v += carryUnsigned(low, ql[i]);
            low += ql[i];
          }
        }
      }
    }
  }
  part[w][l] = v;
  partLow[w][l] = low;
  __syncthreads();
  if (w != 0) {
    return;
  }
  bool isNew = false;
  if (in) {
    for (int o = 1; o < 16; ++o) {
      const uint64_t q = part[o][l];
      if (kind == ACC_SUM_F64) {
        v = static_cast<uint64_t>(__double_as_longlong(__longlong_as_double(static_cast<long long>(v)) +
                                                       __longlong_as_double(static_cast<long long>(q))));
      } else if (kind == ACC_MIN) {
        v = q < v ? q : v;
      } else if (kind == ACC_MAX) {
        v = q > v ? q : v;
      } else {
        v += q;
        if (kind == ACC_SUM_I64_HI) {
          v += carryUnsigned(low, partLow[o][l]);
          low += partLow[o][l];
        }
      }
    }
    uint64_t* g = p.table + static_cast<uint64_t>(key) * p.stride;
    if (storeAll) {
      if (j == A) {
        g[0] = kEmpty;
        g[1] = v;
        isNew = v != kNoRow;
      } else {
        g[p.off[j]] = v;
      }
    } else if (j == A) {
      if (v != kNoRow) {
        isNew = atomicMin(reinterpret_cast<unsigned long long*>(g + 1), static_cast<unsigned long long>(v)) == kNoRow;
      }
    } else if (v != accIdentity(kind)) {
      // (a total equal to the identity changes nothing; keys nobody saw are not touched at all)
      if (kind == ACC_SUM_I64) {
        addPartial128Global(g + p.off[j], v, 0);
      } else {
        applyGlobal(g + p.off[j], kind == ACC_COUNT ? static_cast<int32_t>(ACC_SUM_I64_WRAP) : kind, v, p.counters);
      }
    }
  }
  const uint64_t m = ballot(isNew);
  if (m != 0 && l == __ffsll(static_cast<long long>(m)) - 1) {
    atomicAdd(&p.counters->numNewGroups, static_cast<uint32_t>(popc64(m)));
  }
  publishCountersFromLastBlock(mail, gridDim.x * gridDim.y);  // (wave 0: the only one that got here)
}

// What the host derives from a launch to pick an instantiation.
struct FastSignature {
  int k0 = FK_NONE, k1 = FK_NONE, t0 = FK_NONE, t1 = FK_NONE, numLoads = 0, numAccs = 0;
  uint64_t accLo = 0, accHi = 0, accEx = 0;   // 16 bits per accumulator: 0-3, 4-7, 8-11
  uint32_t ind = 0;  // dictionary-wrapped columns (FastShape::IND)
  uint32_t nul = 0;  // columns with a null bitmap (FastShape::NUL)
  uint32_t lk = 0;   // element types of the loaded columns (FastShape::LK)
  uint32_t msk = 0;  // accumulators with a mask column (FastShape::MSK)
  uint64_t ops = 0;  // what each accumulator does with its operand (FastShape::OPS)
  uint32_t kx = kFastNoExtraKeys;  // kinds of the third and fourth key (FastShape::KX)
  bool operator==(const FastSignature& o) const {
    return k0 == o.k0 && k1 == o.k1 && t0 == o.t0 && t1 == o.t1 && numLoads == o.numLoads &&
        numAccs == o.numAccs && accLo == o.accLo && accHi == o.accHi && accEx == o.accEx && ind == o.ind &&
        nul == o.nul && lk == o.lk && msk == o.msk && ops == o.ops && kx == o.kx;
  }
};

using FastLauncher = void (*)(const FastArgs&, int grid, size_t ldsBytes);

template <typename S>
void launchFast(const FastArgs& fa, int grid, size_t ldsBytes) {
  VX_LAUNCH("k_agg_fast", k_agg_fast<S>, grid, 512, ldsBytes, fa);
}

struct FastEntry {
  FastSignature sig;
  int unroll;
  FastLauncher launch;
};

#define VX_FAST_ENTRY(U, K0, K1, T0, T1, NL, NA, LO, HI)                                      \
  FastEntry {                                                                                 \
    FastSignature{K0, K1, T0, T1, NL, NA, LO, HI, 0, 0, 0}, U,                                      \
        &launchFast<FastShape<U, K0, K1, T0, T1, NL, NA, LO, HI>>                             \
  }

// Ahead-of-time shapes. Accumulators appear in the order buildPlan creates
// them (aliased counts excluded).
//  * TPC-H Q1, fused FilterProject + HashAggregation (TpchQueryBuilder.cpp
//    :203-252): keys l_returnflag, l_linestatus; filter l_shipdate (DATE);
//    loads qty=0, ep=1, disc=2, tax=3; sum(qty), count(*), sum(ep),
//    sum(ep*(1-disc)), sum(ep*(1-disc)*(1+tax)), sum(disc).
//  * BASELINE config 1: k BIGINT; sum(v), count(*).
//  * one INTEGER / BIGINT key with sum + count and an INTEGER filter (Q1-like
//    plans over dictionary-encoded flags).
constexpr uint64_t kQ1Lo = packAccs(accDesc(1, 0), accDesc(0), accDesc(1, 1), accDesc(2, 1, 2));
constexpr uint64_t kQ1Hi = packAccs(accDesc(3, 1, 2, 3), accDesc(1, 2));
constexpr uint64_t kC1Lo = packAccs(accDesc(1, 0), accDesc(0));
// Entry with every FastShape parameter (the log line of VX355_LOG_SHAPES=1 prints them in this order).
#define VX_FAST_ENTRY_X(U, K0, K1, T0, T1, NL, NA, LO, HI, IND, NUL, EX, LK, MSK, OPS)                  \
  FastEntry {                                                                                         \
    FastSignature{K0, K1, T0, T1, NL, NA, LO, HI, EX, IND, NUL, LK, MSK, OPS}, U,                       \
        &launchFast<FastShape<U, K0, K1, T0, T1, NL, NA, LO, HI, IND, NUL, EX, LK, MSK, OPS>>           \
  }
// ... and with a third / fourth key (KX).
#define VX_FAST_ENTRY_K(U, K0, K1, T0, T1, NL, NA, LO, HI, IND, NUL, EX, LK, MSK, OPS, KX)              \
  FastEntry {                                                                                         \
    FastSignature{K0, K1, T0, T1, NL, NA, LO, HI, EX, IND, NUL, LK, MSK, OPS, KX}, U,                   \
        &launchFast<FastShape<U, K0, K1, T0, T1, NL, NA, LO, HI, IND, NUL, EX, LK, MSK, OPS, KX>>       \
  }

const FastEntry kFastTable[] = {
    VX_FAST_ENTRY(2, FK_VIEW, FK_VIEW, FK_I32, FK_NONE, 4, 6, kQ1Lo, kQ1Hi),
    VX_FAST_ENTRY(4, FK_VIEW, FK_VIEW, FK_I32, FK_NONE, 4, 6, kQ1Lo, kQ1Hi),
    VX_FAST_ENTRY(2, FK_I64, FK_NONE, FK_NONE, FK_NONE, 1, 2, kC1Lo, 0),
    VX_FAST_ENTRY(4, FK_I64, FK_NONE, FK_NONE, FK_NONE, 1, 2, kC1Lo, 0),
    VX_FAST_ENTRY(4, FK_I32, FK_NONE, FK_NONE, FK_NONE, 1, 2, kC1Lo, 0),
    VX_FAST_ENTRY(4, FK_I64, FK_NONE, FK_I32, FK_NONE, 1, 2, kC1Lo, 0),
    // the nullable variants the bench lines use (operators of a benchmark live for milliseconds: the
    // background hiprtc compile would never be ready for them):
    //  * config 1 with nulls in v (the reference's *_halfnull benchmarks): sum(v), count(v), count(*)
    VX_FAST_ENTRY_X(4, FK_I64, FK_NONE, FK_NONE, FK_NONE, 1, 3, 0xff00fff0ff01ull, 0x0ull, 0x0u, 0x10u, 0x0ull, 0x0u, 0x0u,
                    0x0ull),
    //  * BASELINE.json's wording of configs[1], "8-column scan, 4-key group-by with 6 aggregates": Q1 with
    //    two more low-cardinality INTEGER keys (l_linenumber, a ship-mode code) and sum(qty), sum(ep),
    //    sum(ep*(1-disc)), avg(qty), avg(disc), count(*): loads qty=0, ep=1, disc=2
    VX_FAST_ENTRY_K(4, FK_VIEW, FK_VIEW, FK_I32, FK_NONE, 3, 5, kQ1Lo, packAccs(accDesc(1, 2)), 0x0u, 0x0u, 0x0ull, 0x0u,
                    0x0u, 0x0ull, packExtraKeys(FK_I32, FK_I32)),
    //  * TPC-H Q1 as Velox really hands it over when the FilterProject is NOT fused: keys and operands
    //    dictionary-wrapped by the filter's selected-row indices, the two projections as flat columns
    //    (bench.py --unfused): a new worker process's first operator must not start on k_agg_lds
    VX_FAST_ENTRY_K(4, FK_VIEW, FK_VIEW, FK_NONE, FK_NONE, 5, 6, 0xff21ff11fff0ff01ull, 0xff41ff31ull, 0x133u, 0x0u,
                    0x0ull, 0x0u, 0x0u, 0x0ull, kFastNoExtraKeys),
    //  * TPC-H Q1 with a nullable l_discount
    VX_FAST_ENTRY_X(4, FK_VIEW, FK_VIEW, FK_I32, FK_NONE, 4, 9, 0xf212ff11fff0ff01ull, 0xff2132103213f210ull, 0x0u, 0x40u,
                    0xff20ull, 0x0u, 0x0u, 0x0ull),
};

// ---- LDS-tiled aggregation for high cardinality (BASELINE config 4) ------------
// One HBM atomic per input row is the ceiling of k_agg_global: the chip retires
// ~20 G atomic_add_f64 per second wherever the table lives (L2, Infinity Cache or
// HBM), so 10^9 rows cost >= 50 ms. This path replaces per-row HBM atomics by
// per-row LDS atomics: rows are radix-partitioned in two passes by
// pid = normalized key / B (B groups fit one workgroup's LDS), then one
// workgroup per partition folds its rows into LDS accumulators and touches each
// HBM group row once. Everything streams: 2 x (read + write) of a compact
// record {key, row, operands} plus one read.
constexpr int kRadixMaxAccs = 3;
constexpr int kRadixMaxBins = 4096;   // per level
constexpr int kRadixKeyBits = 32;     // widest key field of a record
constexpr int kRadixMaskBits = 3;     // = kRadixMaxAccs
// Record word 0 = normalized key : keyBits | row of the chunk : rowBits | accumulator mask : 3,
// keyBits = bits of the table's capacity, rowBits = min(31, 61 - keyBits): a 2^28-group table
// takes chunks of 2^31 rows (BASELINE config 4, 10^9 rows, is ONE chunk: every partition is
// folded once).
inline int radixKeyBits(uint64_t capacity) {
  int b = 1;
  while ((1ULL << b) < capacity) {
    ++b;
  }
  return b;
}
inline int radixRowBits(uint64_t capacity) { return std::min(31, 64 - kRadixMaskBits - radixKeyBits(capacity)); }

struct RadixArgs {
  AggArgs a;
  int32_t numVals;     // operand words per record (accumulators other than counts)
  int32_t recWords;    // 1 + numVals
  int32_t shiftB;      // log2(groups per partition): pid = key >> shiftB
  int32_t shift2;      // level-1 bin = pid >> shift2 (0 = single level)
  int32_t numBins;     // level-1 bins
  int32_t keyBits;     // layout of record word 0
  int32_t rowBits;
  // Open-addressing tables (normalized-key mode, sparse keys): the records are partitioned by the
  // group's HOME SLOT = twang_mix64(normalized key) & slotMask instead of by the key itself, and
  // carry the full key in their last word (k_rp_aggregate_hashed).
  int32_t hashed;
  int32_t pad0;
  uint64_t slotMask;
  int32_t valIdx[kRadixMaxAccs];  // operand word of accumulator j, -1 for counts
  int32_t accOfVal[kRadixMaxAccs];  // inverse: accumulator of operand word q
  int64_t tileRows;    // rows per workgroup tile
  int64_t numTiles;
  uint32_t* hist;      // [bin][tile]
  const uint64_t* offsets;
  uint64_t* recs;      // output records of this pass
  // Optimistic level 1 (no counting pass over the keys): bin b owns the region
  // [binFirst[b], binFirst[b] + binCap) of 'recs' and a cursor; a sub-tile's run of records claims
  // its place with one atomic. A bin that outgrows its region sets *binOverflow and the host redoes
  // the level with the exact two-pass form.
  const uint64_t* binFirst;
  uint32_t* binCursor;
  uint32_t* binOverflow;
  uint64_t binCap;
  uint64_t crCap;      // compact records: record capacity of 'recs' (see recLoad); rowBits is 0 then
};
static_assert(sizeof(RadixArgs) <= 4096, "kernel arguments are limited to 4 KB");

// KW = 8 / 4: a single flat BIGINT / INTEGER key without nulls and no fused filter
// (loads issued ahead of their use, kRadixUnroll rows per lane in flight);
// KW = 0: any key set the ABI admits.
constexpr int kRadixUnroll = 8;

// Read-once streams of the radix passes (input columns, records): nontemporal under -DVX355_RP_NT_LOADS.
#ifdef VX355_RP_NT_LOADS
#define VX355_RP_LOAD(p) __builtin_nontemporal_load(p)
#else
#define VX355_RP_LOAD(p) (*(p))
#endif
typedef unsigned long long RpU64x2 __attribute__((ext_vector_type(2)));

template <int KW>
__device__ inline int64_t rpLoadKey(const RadixArgs& r, int64_t row) {
  if constexpr (KW == 8) {
    return VX355_RP_LOAD(static_cast<const int64_t*>(r.a.keys[0].col.values) + row);
  } else if constexpr (KW == 4) {
    return VX355_RP_LOAD(static_cast<const int32_t*>(r.a.keys[0].col.values) + row);
  } else {
    return 0;
  }
}

template <int KW>
__device__ inline int rpKey(const RadixArgs& r, int64_t row, int64_t raw, uint64_t* key) {
  const AggArgs& a = r.a;
  if constexpr (KW != 0) {
    const KeyRange& kr = a.keys[0].range;
    if (raw >= kr.min && raw <= kr.max) {
      *key = static_cast<uint64_t>(raw) - static_cast<uint64_t>(kr.min) + 1;
      return 0;
    }
    return normalizedKey(a, row, key);  // out of range: statistics + deferral
  } else {
    return (a.numTerms && !evalFilter(a.terms, a.numTerms, row)) ? 1 : normalizedKey(a, row, key);
  }
}

// Level 1, count: histogram of the level-1 bin over the rows of each tile that
// pass the filter and map into the current key ranges (others: deferred list).
template <int KW>
__global__ __launch_bounds__(1024) void k_rp_count1(RadixArgs r) {
  __shared__ uint32_t hist[kRadixMaxBins];
  const AggArgs& a = r.a;
  const int shift = r.shiftB + r.shift2;
  for (int64_t tile = blockIdx.x; tile < r.numTiles; tile += gridDim.x) {
    for (int i = threadIdx.x; i < r.numBins; i += blockDim.x) {
      hist[i] = 0;
    }
    blockSync();
    const int64_t begin = tile * r.tileRows;
    const int64_t end = begin + r.tileRows < a.numRows ? begin + r.tileRows : a.numRows;
    for (int64_t base = begin; base < end; base += kRadixUnroll * 1024) {
      int64_t raw[kRadixUnroll];
#pragma unroll
      for (int u = 0; u < kRadixUnroll; ++u) {
        const int64_t row = base + u * 1024 + threadIdx.x;
        raw[u] = row < end ? rpLoadKey<KW>(r, row) : 0;
      }
#pragma unroll
      for (int u = 0; u < kRadixUnroll; ++u) {
        const int64_t row = base + u * 1024 + threadIdx.x;
        bool defer = false;
        if (row < end) {
          uint64_t key;
          const int st = rpKey<KW>(r, row, raw[u], &key);
          if (st == 0) {
            const uint64_t part = r.hashed ? (twangMix64(key) & r.slotMask) : key;
            atomicAdd(&hist[part >> shift], 1u);
          } else if (st == 2) {
            defer = true;
          }
        }
        deferRow(a, defer, static_cast<int32_t>(row));
      }
    }
    blockSync();
    for (int i = threadIdx.x; i < r.numBins; i += blockDim.x) {
      r.hist[static_cast<int64_t>(i) * r.numTiles + tile] = hist[i];
    }
    blockSync();
  }
}

template <int W>
__device__ inline void rpStore(uint64_t* dst, const uint64_t* w) {
  if constexpr (W == 2) {
    *reinterpret_cast<ulonglong2*>(dst) = make_ulonglong2(w[0], w[1]);
  } else if constexpr (W == 4) {
    reinterpret_cast<ulonglong2*>(dst)[0] = make_ulonglong2(w[0], w[1]);
    reinterpret_cast<ulonglong2*>(dst)[1] = make_ulonglong2(w[2], w[3]);
  } else {
#pragma unroll
    for (int i = 0; i < W; ++i) {
      dst[i] = w[i];
    }
  }
}

template <int W>
__device__ inline void rpLoad(const uint64_t* src, uint64_t* w) {
  if constexpr (W == 2) {
    const RpU64x2 v = VX355_RP_LOAD(reinterpret_cast<const RpU64x2*>(src));
    w[0] = v.x;
    w[1] = v.y;
  } else if constexpr (W == 4) {
    const RpU64x2 v0 = VX355_RP_LOAD(reinterpret_cast<const RpU64x2*>(src));
    const RpU64x2 v1 = VX355_RP_LOAD(reinterpret_cast<const RpU64x2*>(src) + 1);
    w[0] = v0.x;
    w[1] = v0.y;
    w[2] = v1.x;
    w[3] = v1.y;
  } else {
#pragma unroll
    for (int i = 0; i < W; ++i) {
      w[i] = VX355_RP_LOAD(src + i);
    }
  }
}

// Compact records (CR; round 5): when nobody asked for first-seen order, the row number need not travel and
// word 0 = {key : keyBits | accumulator mask : 3} fits 32 bits - a 12-byte record {word 0, operand lo, operand hi}
// instead of 16, record i at byte 12 i of the buffer (three dwords, 4-byte aligned: one dwordx3 access per
// lane, a wave covers 768 consecutive bytes). One operand, direct-index tables, optimistic levels only
// (launchRadix). In LDS and in registers a record stays two 64-bit words; only the HBM side is narrower:
// 28 + 24 + 12 instead of 32 + 32 + 16 bytes per row through the three passes. 'cap' (the buffer's record
// capacity) is only the "compact" flag of the host side.
struct __attribute__((aligned(4))) Rec12 {
  uint32_t word0, lo, hi;
};
__device__ inline uint32_t crWord0(const uint64_t* base, uint64_t i) {
  return reinterpret_cast<const Rec12*>(base)[i].word0;
}
template <int W, bool CR>
__device__ inline void recLoad(const uint64_t* base, uint64_t cap, uint64_t i, uint64_t* w) {
  if constexpr (CR) {
    static_assert(W == 2, "compact records carry one operand");
    (void)cap;
    const Rec12* p = reinterpret_cast<const Rec12*>(base) + i;
    const uint32_t w0 = VX355_RP_LOAD(&p->word0), lo = VX355_RP_LOAD(&p->lo), hi = VX355_RP_LOAD(&p->hi);
    w[0] = w0;   // (zero extension: no use of the loaded value, the load stays in flight)
    w[1] = static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32);
  } else {
    rpLoad<W>(base + i * W, w);
  }
}
template <int W, bool CR>
__device__ inline void recStore(uint64_t* base, uint64_t cap, uint64_t i, const uint64_t* w) {
  if constexpr (CR) {
    (void)cap;
    Rec12* p = reinterpret_cast<Rec12*>(base) + i;
    p->word0 = static_cast<uint32_t>(w[0]);
    p->lo = static_cast<uint32_t>(w[1]);
    p->hi = static_cast<uint32_t>(w[1] >> 32);
  } else {
    rpStore<W>(base + i * W, w);
  }
}

// Level 1, scatter: the same rows, as records of W words, to their level-1
// bucket. FLATV: every operand is a flat 8-byte column without nulls or mask,
// loaded ahead like the key.
template <int KW, int W, bool FLATV>
__global__ __launch_bounds__(1024) void k_rp_scatter1(RadixArgs r) {
  // Output position = base[bin] (u64, read only) + rank from a 32-bit LDS
  // atomic (64-bit LDS atomics run at half rate).
  __shared__ unsigned long long binBase[kRadixMaxBins];
  __shared__ uint32_t cursor[kRadixMaxBins];
  const AggArgs& a = r.a;
  const int shift = r.shiftB + r.shift2;
  for (int64_t tile = blockIdx.x; tile < r.numTiles; tile += gridDim.x) {
    for (int i = threadIdx.x; i < r.numBins; i += blockDim.x) {
      binBase[i] = r.offsets[static_cast<int64_t>(i) * r.numTiles + tile];
      cursor[i] = 0;
    }
    blockSync();
    const int64_t begin = tile * r.tileRows;
    const int64_t end = begin + r.tileRows < a.numRows ? begin + r.tileRows : a.numRows;
    for (int64_t base = begin; base < end; base += kRadixUnroll * 1024) {
      int64_t raw[kRadixUnroll];
      uint64_t vals[kRadixUnroll][W];
#pragma unroll
      for (int u = 0; u < kRadixUnroll; ++u) {
        const int64_t row = base + u * 1024 + threadIdx.x;
        raw[u] = row < end ? rpLoadKey<KW>(r, row) : 0;
        if constexpr (FLATV) {
#pragma unroll
          for (int q = 1; q < W; ++q) {
            vals[u][q] = row < end ? VX355_RP_LOAD(static_cast<const uint64_t*>(a.accs[r.accOfVal[q - 1]].in.values) + row) : 0;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kRadixUnroll; ++u) {
        const int64_t row = base + u * 1024 + threadIdx.x;
        if (row >= end) {
          continue;
        }
        uint64_t key;
        if (rpKey<KW>(r, row, raw[u], &key) != 0) {
          continue;
        }
        uint64_t mask = 0;
        if constexpr (FLATV) {
          mask = (1ULL << a.numAccs) - 1;
#pragma unroll
          for (int q = 1; q < W; ++q) {
            const AccArg& acc = a.accs[r.accOfVal[q - 1]];
            if (acc.kind == ACC_MIN || acc.kind == ACC_MAX) {
              vals[u][q] = acc.inIsInt ? int64ToOrdered(static_cast<int64_t>(vals[u][q]))
                                       : doubleToOrdered(__longlong_as_double(static_cast<long long>(vals[u][q])));
            }
          }
        } else {
#pragma unroll
          for (int q = 1; q < W; ++q) {
            const int j = r.accOfVal[q - 1];
            vals[u][q] = 0;
            if (accInput(a, a.accs[j], row, &vals[u][q])) {
              mask |= 1ULL << j;
            }
          }
          for (int j = 0; j < a.numAccs; ++j) {
            uint64_t one;
            if (r.valIdx[j] < 0 && accInput(a, a.accs[j], row, &one)) {
              mask |= 1ULL << j;
            }
          }
        }
        vals[u][0] = key | (static_cast<uint64_t>(row) << r.keyBits) | (mask << (r.keyBits + r.rowBits));
        const uint32_t bin = static_cast<uint32_t>(key >> shift);
        const unsigned long long pos = binBase[bin] + atomicAdd(&cursor[bin], 1u);
        rpStore<W>(r.recs + pos * W, vals[u]);
      }
    }
    blockSync();
  }
}

// ---- scatter with an LDS sort in front (tools/scatter_bench.hip, profiles/r03_scatter_bench.txt) ----
// A lane that stores its 16-byte record straight to base[bin] + cursor makes every store
// instruction touch 64 unrelated lines: 2.9 TB/s at 191 bins, 2.7 at 382 (read + write), against
// 4.8 TB/s for a plain copy. Counting-sorting every sub-tile of kSortSub records by bin inside LDS
// first (histogram, scan, placement) turns the records of a bin into a run of consecutive lanes
// storing to consecutive addresses: 4.5 TB/s at 191 bins, 4.0 at 382 with sub-tiles of 4096
// records; the longer the runs the better, so a sub-tile takes what the LDS holds: 8192 16-byte
// records (128 KB), one workgroup of 1024 lanes per CU, 8 rows per lane.
constexpr int kSortBins = 1024;    // widest fan-out of the sorted scatters
constexpr int kSortThreads = 1024;
template <int W>
struct SortLds {
  static constexpr int kSub = (W <= 2 ? 8192 : 4096);   // records per sub-tile: <= 128 KB of LDS
  static constexpr int kRounds = kSub / kSortThreads;   // rows per lane and sub-tile
  unsigned long long binBase[kSortBins];  // next free record of the bin inside this tile's range
  uint32_t cnt[kSortBins];                // sub-tile histogram, then placement cursor
  uint32_t start[kSortBins];              // sub-tile exclusive scan
  uint64_t recs[kSub * W];
  uint32_t waveTotals[kSortThreads / 64];
};

// The sub-tile's records (rec[u], bin[u] = 0xffffffff: none) leave for their bins. Called by all
// 1024 lanes; cnt[] must be zero on entry and is zero again on return.
// binOfWord0: bin of a record from its first word (the write-out pass recomputes it instead of
// keeping a bin per staged record in LDS).
struct NoReserve {};
// reserve (optional): called by the lane of every non-empty bin with (bin, records of the sub-tile);
// returns the first record position of that run in 'out', or ~0 to drop the run — the optimistic
// level 2 claims its space from a global cursor per partition instead of a counted offset.
// First half: histogram, scan, (reserve,) placement of the records in LDS, sorted by bin. On return
// the lanes' record registers are dead - the caller issues the NEXT sub-tile's loads here, so that
// they are in flight while rpSortedWrite streams this sub-tile out (HBM reads overlap HBM writes;
// with one workgroup per CU nothing else would fill the read side during the write-out).
template <int W, int R, typename ReserveFn = NoReserve>
__device__ inline void rpSortedPlace(SortLds<W>& l, int numBins, const uint64_t (&rec)[R][W], const uint32_t (&bin)[R],
                                     ReserveFn&& reserve = NoReserve{}) {
  constexpr bool kReserves = !std::is_same<typename std::decay<ReserveFn>::type, NoReserve>::value;
  const int tid = threadIdx.x;
#pragma unroll
  for (int u = 0; u < R; ++u) {
    if (bin[u] != 0xffffffffu) {
      atomicAdd(&l.cnt[bin[u]], 1u);
    }
  }
  blockSync();
  // exclusive scan of the histogram: one bin per lane
  const uint32_t mine = tid < numBins ? l.cnt[tid] : 0;
  uint32_t incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off, kWave);
    if (lane() >= off) {
      incl += o;
    }
  }
  if (lane() == 63) {
    l.waveTotals[tid >> 6] = incl;
  }
  blockSync();
  uint32_t run = incl - mine;
  for (int w = 0; w < (tid >> 6); ++w) {
    run += l.waveTotals[w];
  }
  if (tid < numBins) {
    l.start[tid] = run;
    l.cnt[tid] = run;  // placement cursor
    if constexpr (kReserves) {
      if (mine != 0) {
        l.binBase[tid] = reserve(static_cast<uint32_t>(tid), mine);
      }
    }
  }
  blockSync();
#pragma unroll
  for (int u = 0; u < R; ++u) {
    if (bin[u] != 0xffffffffu) {
      const uint32_t pos = atomicAdd(&l.cnt[bin[u]], 1u);
      rpStore<W>(l.recs + static_cast<size_t>(pos) * W, rec[u]);
    }
  }
  blockSync();
}

// Second half: the sorted sub-tile leaves for its bins; cnt[] is zero again on return.
template <int W, bool kReserves, typename BinFn, bool CR = false>
__device__ inline void rpSortedWrite(SortLds<W>& l, int numBins, uint64_t* out, BinFn&& binOfWord0, uint64_t crCap = 0) {
  const int tid = threadIdx.x;
  uint32_t total = 0;
#pragma unroll
  for (int w = 0; w < kSortThreads / 64; ++w) {
    total += l.waveTotals[w];
  }
  for (uint32_t i = tid; i < total; i += kSortThreads) {
    uint64_t w[W];
    rpLoad<W>(l.recs + static_cast<size_t>(i) * W, w);
    const uint32_t b = binOfWord0(w[0]);
    if (kReserves && l.binBase[b] == ~0ULL) {
      continue;  // the partition's optimistic region is full: the host redoes the level exactly
    }
    recStore<W, CR>(out, crCap, l.binBase[b] + (i - l.start[b]), w);
  }
  blockSync();
  if (tid < numBins) {
    if constexpr (!kReserves) {
      l.binBase[tid] += l.cnt[tid] - l.start[tid];
    }
    l.cnt[tid] = 0;
  }
  blockSync();
}

template <int W, int R, typename BinFn, typename ReserveFn = NoReserve>
__device__ inline void rpSortedEmit(SortLds<W>& l, int numBins, const uint64_t (&rec)[R][W], const uint32_t (&bin)[R],
                                    uint64_t* out, BinFn&& binOfWord0, ReserveFn&& reserve = NoReserve{}) {
  constexpr bool kReserves = !std::is_same<typename std::decay<ReserveFn>::type, NoReserve>::value;
  rpSortedPlace<W, R>(l, numBins, rec, bin, reserve);
  rpSortedWrite<W, kReserves>(l, numBins, out, binOfWord0);
}

// Level 1 with sorted sub-tiles (numBins <= kSortBins); same records as k_rp_scatter1.
// HASHED: word 0 carries the home slot instead of the key, the last word the full key.
template <int KW, int W, bool FLATV, bool HASHED, bool OPT = false, bool CR = false, bool KR = false>
__global__ __launch_bounds__(kSortThreads) void k_rp_scatter1_sorted(RadixArgs r) {
  __shared__ SortLds<W> l;
  constexpr int R = SortLds<W>::kRounds;
  static_assert(!KR || (HASHED && FLATV && OPT && W == 2 && !CR), "keyed records: see hashRecLoad");
  constexpr int V = W - (HASHED && !KR ? 1 : 0);   // word 0 + operand words
  const AggArgs& a = r.a;
  const int shift = r.shiftB + r.shift2;
  for (int i = threadIdx.x; i < kSortBins; i += kSortThreads) {
    l.cnt[i] = 0;
  }
  for (int64_t tile = blockIdx.x; tile < r.numTiles; tile += gridDim.x) {
    if constexpr (!OPT) {
      for (int i = threadIdx.x; i < r.numBins; i += kSortThreads) {
        l.binBase[i] = r.offsets[static_cast<int64_t>(i) * r.numTiles + tile];
      }
    }
    blockSync();
    const int64_t begin = tile * r.tileRows;
    const int64_t end = begin + r.tileRows < a.numRows ? begin + r.tileRows : a.numRows;
    int64_t raw[R];
    uint64_t vals[R][W];
    // the sub-tile's key and operand loads (clamped rows: every lane loads, the results of rows >= end
    // are ignored); called for sub-tile i + 1 between the two halves of sub-tile i's sorted emit
    auto loadSub = [&](int64_t base) {
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int64_t row = base + u * kSortThreads + threadIdx.x;
        const int64_t at = row < end ? row : end - 1;
        raw[u] = rpLoadKey<KW>(r, at);
        if constexpr (FLATV) {
#pragma unroll
          for (int q = 1; q < V; ++q) {
            vals[u][q] = VX355_RP_LOAD(static_cast<const uint64_t*>(a.accs[r.accOfVal[q - 1]].in.values) + at);
          }
        }
      }
    };
    if (begin < end) {
      loadSub(begin);
    }
    for (int64_t base = begin; base < end; base += SortLds<W>::kSub) {
      uint32_t bin[R];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int64_t row = base + u * kSortThreads + threadIdx.x;
        bin[u] = 0xffffffffu;
        uint64_t key;
        if (row >= end || rpKey<KW>(r, row, raw[u], &key) != 0) {
          continue;
        }
        uint64_t mask = 0;
        if constexpr (FLATV) {
          mask = (1ULL << a.numAccs) - 1;
#pragma unroll
          for (int q = 1; q < V; ++q) {
            const AccArg& acc = a.accs[r.accOfVal[q - 1]];
            if (acc.kind == ACC_MIN || acc.kind == ACC_MAX) {
              vals[u][q] = acc.inIsInt ? int64ToOrdered(static_cast<int64_t>(vals[u][q]))
                                       : doubleToOrdered(__longlong_as_double(static_cast<long long>(vals[u][q])));
            }
          }
        } else {
#pragma unroll
          for (int q = 1; q < V; ++q) {
            const int j = r.accOfVal[q - 1];
            vals[u][q] = 0;
            if (accInput(a, a.accs[j], row, &vals[u][q])) {
              mask |= 1ULL << j;
            }
          }
          for (int j = 0; j < a.numAccs; ++j) {
            uint64_t one;
            if (r.valIdx[j] < 0 && accInput(a, a.accs[j], row, &one)) {
              mask |= 1ULL << j;
            }
          }
        }
        uint64_t part = key;
        if constexpr (HASHED) {
          part = twangMix64(key) & r.slotMask;
          if constexpr (!KR) {
            vals[u][W - 1] = key;
          }
        }
        // (compact records: no row, rowBits = 0 - the mask sits right behind the key; keyed records: the key alone)
        vals[u][0] = KR ? key
                        : (part | (CR ? 0 : (static_cast<uint64_t>(row) << r.keyBits)) | (mask << (r.keyBits + r.rowBits)));
        bin[u] = static_cast<uint32_t>(part >> shift);
      }
      const uint64_t keyMask = (1ULL << r.keyBits) - 1;
      const int64_t next = base + SortLds<W>::kSub;
      if constexpr (OPT) {
        rpSortedPlace<W, R>(l, r.numBins, vals, bin, [&](uint32_t b, uint32_t count) -> unsigned long long {
          const uint32_t at = atomicAdd(&r.binCursor[b], count);
          if (static_cast<uint64_t>(at) + count > r.binCap) {
            *r.binOverflow = 1;
            return ~0ULL;
          }
          return r.binFirst[b] + at;
        });
        if (next < end) {
          loadSub(next);
        }
        auto binOf = [&](uint64_t w0) {
          return static_cast<uint32_t>(((KR ? (twangMix64(w0) & r.slotMask) : w0) & keyMask) >> shift);
        };
        rpSortedWrite<W, true, decltype(binOf)&, CR>(l, r.numBins, r.recs, binOf, r.crCap);
      } else {
        rpSortedPlace<W, R>(l, r.numBins, vals, bin);
        if (next < end) {
          loadSub(next);
        }
        rpSortedWrite<W, false>(l, r.numBins, r.recs,
                                [&](uint64_t w0) { return static_cast<uint32_t>((w0 & keyMask) >> shift); });
      }
    }
  }
}

// Level 2 works on records; tiles never straddle level-1 buckets.
struct RadixTile {
  uint64_t begin;   // first record
  uint32_t count;   // records in the tile
  uint32_t cell;    // index of (this tile, bin 0) in the level-2 histogram; bins are 'stride' cells apart
  uint32_t stride;  // tiles of the bucket
  uint32_t pad;
};

// Builds the level-2 tile table and the partition -> histogram cell map from
// the level-1 offsets, on device (no host round trip between the levels).
constexpr uint32_t kDeadBinCursor = 0xffffff00u;

// Where level 1 put bin b: counted offsets, or (optimistic level 1) the bin's region and cursor.
struct Level1Bins {
  const uint64_t* offsets1;
  int64_t numTiles1;
  const uint64_t* binFirst;   // non-null: optimistic level 1
  const uint32_t* binCursor;
  __device__ uint64_t first(int b) const {
    return binFirst ? binFirst[b] : offsets1[static_cast<int64_t>(b) * numTiles1];
  }
  __device__ uint64_t count(int b) const {
    if (binFirst) {
      // (a bin outside the observed key range has no region: its cursor starts at kDeadBinCursor
      // and is never advanced without raising the overflow flag)
      const uint32_t c = binCursor[b];
      return c >= kDeadBinCursor ? 0 : c;
    }
    return offsets1[static_cast<int64_t>(b + 1) * numTiles1] - offsets1[static_cast<int64_t>(b) * numTiles1];
  }
};

__global__ __launch_bounds__(1024) void k_rp_tiles(Level1Bins l1, int32_t numBins1,
                                                    int32_t numBins2, int32_t shift2, uint32_t tileRecs,
                                                    RadixTile* tiles, uint32_t* numTiles2, uint32_t* partCell,
                                                    int64_t numParts) {
  // (every workgroup derives the same tile starts, then takes its share of the two tables)
  __shared__ uint32_t tileStart[kRadixMaxBins + 1];
  for (int b = threadIdx.x; b < numBins1; b += blockDim.x) {
    const uint64_t count = l1.count(b);
    tileStart[b] = static_cast<uint32_t>((count + tileRecs - 1) / tileRecs);
  }
  blockSync();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int b = 0; b < numBins1; ++b) {
      const uint32_t n = tileStart[b];
      tileStart[b] = run;
      run += n;
    }
    tileStart[numBins1] = run;
    if (blockIdx.x == 0) {
      *numTiles2 = run;
    }
  }
  blockSync();
  const int64_t lanes = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t me = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint32_t totalTiles = tileStart[numBins1];
  for (int64_t at = me; at < totalTiles; at += lanes) {
    // the bucket of tile 'at': the last one that starts at or before it and is not empty
    int lo = 0, hi = numBins1 - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tileStart[mid] <= at) {
        lo = mid;
      } else {
        hi = mid - 1;
      }
    }
    const int b = lo;
    const uint32_t j = static_cast<uint32_t>(at) - tileStart[b];
    const uint64_t count = l1.count(b);
    RadixTile t;
    t.begin = l1.first(b) + static_cast<uint64_t>(j) * tileRecs;
    const uint64_t left = count - static_cast<uint64_t>(j) * tileRecs;
    t.count = static_cast<uint32_t>(left < tileRecs ? left : tileRecs);
    t.cell = static_cast<uint32_t>(numBins2) * tileStart[b] + j;
    t.stride = tileStart[b + 1] - tileStart[b];
    t.pad = 0;
    tiles[at] = t;
  }
  for (int64_t p = me; p <= numParts; p += lanes) {
    const int64_t b1 = p >> shift2;
    const uint32_t b2 = static_cast<uint32_t>(p & (numBins2 - 1));
    uint32_t cell;
    if (b1 >= numBins1) {
      cell = static_cast<uint32_t>(numBins2) * tileStart[numBins1];
    } else {
      cell = static_cast<uint32_t>(numBins2) * tileStart[b1] + b2 * (tileStart[b1 + 1] - tileStart[b1]);
    }
    partCell[p] = cell;
  }
}

struct Radix2Args {
  const uint64_t* in;   // records grouped by level-1 bucket
  uint64_t* out;
  const RadixTile* tiles;
  const uint32_t* numTiles;
  int32_t recWords;
  int32_t shiftB;
  int32_t numBins;      // level-2 bins (power of two)
  int32_t pad;
  uint32_t* hist;
  const uint64_t* offsets;
};

__global__ __launch_bounds__(1024) void k_rp_count2(Radix2Args r) {
  __shared__ uint32_t hist[kRadixMaxBins];
  const uint32_t numTiles = *r.numTiles;
  const uint32_t binMask = static_cast<uint32_t>(r.numBins - 1);
  for (uint32_t t = blockIdx.x; t < numTiles; t += gridDim.x) {
    const RadixTile tile = r.tiles[t];
    for (int i = threadIdx.x; i < r.numBins; i += blockDim.x) {
      hist[i] = 0;
    }
    blockSync();
    for (uint32_t i = threadIdx.x; i < tile.count; i += blockDim.x) {
      const uint32_t key = static_cast<uint32_t>(r.in[(tile.begin + i) * r.recWords]);
      atomicAdd(&hist[(key >> r.shiftB) & binMask], 1u);
    }
    blockSync();
    for (int i = threadIdx.x; i < r.numBins; i += blockDim.x) {
      r.hist[tile.cell + static_cast<uint64_t>(i) * tile.stride] = hist[i];
    }
    blockSync();
  }
}

template <int W>
__global__ __launch_bounds__(1024) void k_rp_scatter2(Radix2Args r) {
  __shared__ unsigned long long binBase[kRadixMaxBins];
  __shared__ uint32_t cursor[kRadixMaxBins];
  const uint32_t numTiles = *r.numTiles;
  const uint32_t binMask = static_cast<uint32_t>(r.numBins - 1);
  for (uint32_t t = blockIdx.x; t < numTiles; t += gridDim.x) {
    const RadixTile tile = r.tiles[t];
    for (int i = threadIdx.x; i < r.numBins; i += blockDim.x) {
      binBase[i] = r.offsets[tile.cell + static_cast<uint64_t>(i) * tile.stride];
      cursor[i] = 0;
    }
    blockSync();
    for (uint32_t base = 0; base < tile.count; base += kRadixUnroll * 1024) {
      uint64_t w[kRadixUnroll][W];
#pragma unroll
      for (int u = 0; u < kRadixUnroll; ++u) {
        // (clamped, unconditional: a load inside `if (i < count)` is fused with its use by the compiler
        // and every one of the kRadixUnroll loads then waits for the previous one - s_waitcnt vmcnt(0)
        // behind each global_load in the ISA; issued back to back they overlap)
        const uint32_t i = base + u * 1024 + threadIdx.x;
        rpLoad<W>(r.in + (tile.begin + (i < tile.count ? i : tile.count - 1)) * W, w[u]);
      }
#pragma unroll
      for (int u = 0; u < kRadixUnroll; ++u) {
        const uint32_t i = base + u * 1024 + threadIdx.x;
        if (i < tile.count) {
          const uint32_t bin = (static_cast<uint32_t>(w[u][0]) >> r.shiftB) & binMask;
          const unsigned long long pos = binBase[bin] + atomicAdd(&cursor[bin], 1u);
          rpStore<W>(r.out + pos * W, w[u]);
        }
      }
    }
    blockSync();
  }
}

// Level 2 with sorted sub-tiles (numBins <= kSortBins).
template <int W>
__device__ inline void rpScatter2SortedBody(const Radix2Args& r) {
  __shared__ SortLds<W> l;
  constexpr int R = SortLds<W>::kRounds;
  const uint32_t numTiles = *r.numTiles;
  const uint32_t binMask = static_cast<uint32_t>(r.numBins - 1);
  for (int i = threadIdx.x; i < kSortBins; i += kSortThreads) {
    l.cnt[i] = 0;
  }
  for (uint32_t t = blockIdx.x; t < numTiles; t += gridDim.x) {
    const RadixTile tile = r.tiles[t];
    for (int i = threadIdx.x; i < r.numBins; i += kSortThreads) {
      l.binBase[i] = r.offsets[tile.cell + static_cast<uint64_t>(i) * tile.stride];
    }
    blockSync();
    uint64_t w[R][W];
    // the sub-tile's records (clamped, unconditional loads: see k_rp_scatter2)
    auto loadSub = [&](uint32_t base) {
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const uint32_t i = base + u * kSortThreads + threadIdx.x;
        rpLoad<W>(r.in + (tile.begin + (i < tile.count ? i : tile.count - 1)) * W, w[u]);
      }
    };
    // (loading sub-tile i + 1 between the halves of sub-tile i's emit, as level 1 does, made this pass
    // SLOWER: 7.27 -> 7.60 ms dense, 15.0 -> 16.6 ms sparse keys at 10^9 rows - its reads and writes fall
    // into the same bucket's memory; profiles/r05_c4_pass_bytes.md)
    for (uint32_t base = 0; base < tile.count; base += SortLds<W>::kSub) {
      loadSub(base);
      uint32_t bin[R];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const uint32_t i = base + u * kSortThreads + threadIdx.x;
        bin[u] = i < tile.count ? ((static_cast<uint32_t>(w[u][0]) >> r.shiftB) & binMask) : 0xffffffffu;
      }
      rpSortedPlace<W, R>(l, r.numBins, w, bin);
      rpSortedWrite<W, false>(l, r.numBins, r.out,
                              [&](uint64_t w0) { return (static_cast<uint32_t>(w0) >> r.shiftB) & binMask; });
    }
  }
}

template <int W>
__global__ __launch_bounds__(kSortThreads) void k_rp_scatter2_sorted(Radix2Args r) {
  rpScatter2SortedBody<W>(r);
}

// The same pass over the packed 8-byte entries of the first-seen sort (see k_fs_pack), under a name of
// its own: profiles and counters of the radix path's level 2 stay its own.
__global__ __launch_bounds__(kSortThreads) void k_fs_scatter(Radix2Args r) {
  rpScatter2SortedBody<1>(r);
}

// ---- optimistic level 2: no counting pass -----------------------------------------------------
// k_rp_count2 reads every record of level 1 once more (16 GB at 10^9 rows) only to size the
// partitions exactly. Level 1 already knows every bucket's record count; when keys are spread
// evenly inside a bucket each of its 2^shift2 partitions holds about 1 / 2^shift2 of them. So the
// partitions of bucket b get regions of cap_b = 1.5 x that share + 256 records, a sub-tile claims
// the space of each of its runs with ONE atomic on the partition's cursor, and a partition that
// outgrows its region raises a flag: the host then redoes the level with the exact passes
// (count2 + scan + scatter2). Uniform keys never raise it (the share's standard deviation is
// ~1 % of it); skewed keys pay one wasted pass.
struct Radix2OptArgs {
  const uint64_t* in;
  uint64_t* out;
  const RadixTile* tiles;
  const uint32_t* numTiles;
  int32_t shiftB;
  int32_t shift2;
  int32_t numBins;          // 2^shift2
  int32_t keyBits;          // key field of record word 0
  const uint64_t* partBase; // [numParts + 1]
  const uint32_t* bucketCap;  // region size of every partition of bucket b
  uint32_t* partCount;      // cursors, zero on entry
  uint32_t* overflow;       // set when a region is full
  uint64_t crCapIn, crCapOut;  // compact records: record capacities of 'in' and 'out'
  uint64_t slotMask;           // keyed records (KR): home slot = twangMix64(word 0) & slotMask
};

// partBase / bucketCap from the level-1 bucket sizes (one workgroup; buckets <= kRadixMaxBins).
// Every partition's region holds 1.5 x the even share of the FULLEST bucket + 256 records — a
// bucket at the edge of the live key range has few live partitions, each as full as those of a
// full bucket — but never more than its bucket holds.
__global__ __launch_bounds__(1024) void k_rp_layout2(Level1Bins l1, int32_t numBins1,
                                                      int32_t shift2, int64_t numParts, uint64_t* partBase,
                                                      uint32_t* bucketCap, uint64_t* totalOut) {
  __shared__ unsigned long long bucketBase[kRadixMaxBins + 1];
  __shared__ unsigned long long maxCount;
  const int32_t bins2 = 1 << shift2;
  if (threadIdx.x == 0) {
    maxCount = 0;
  }
  blockSync();
  for (int b = threadIdx.x; b < numBins1; b += blockDim.x) {
    const unsigned long long count = l1.count(b);
    atomicMax(&maxCount, count);
  }
  blockSync();
  const uint64_t share = (maxCount + bins2 - 1) / bins2;
  const uint64_t cap = share + share / 2 + 256;
  for (int b = threadIdx.x; b < numBins1; b += blockDim.x) {
    const uint64_t count = l1.count(b);
    bucketCap[b] = static_cast<uint32_t>(count < cap ? count : cap);
  }
  blockSync();
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int b = 0; b < numBins1; ++b) {
      bucketBase[b] = run;
      run += static_cast<unsigned long long>(bucketCap[b]) * bins2;
    }
    bucketBase[numBins1] = run;
    *totalOut = run;
  }
  blockSync();
  for (int64_t p = threadIdx.x; p <= numParts; p += blockDim.x) {
    const int64_t b = p >> shift2;
    partBase[p] = b >= numBins1 ? bucketBase[numBins1]
                                : bucketBase[b] + static_cast<uint64_t>(p & (bins2 - 1)) * bucketCap[b];
  }
}

template <int W, bool CR = false, bool KR = false>
__global__ __launch_bounds__(kSortThreads) void k_rp_scatter2_opt(Radix2OptArgs r) {
  __shared__ SortLds<W> l;
  constexpr int R = SortLds<W>::kRounds;
  const uint32_t numTiles = *r.numTiles;
  const uint32_t binMask = static_cast<uint32_t>(r.numBins - 1);
  for (int i = threadIdx.x; i < kSortBins; i += kSortThreads) {
    l.cnt[i] = 0;
  }
  blockSync();
  for (uint32_t t = blockIdx.x; t < numTiles; t += gridDim.x) {
    const RadixTile tile = r.tiles[t];
    // every record of a level-2 tile belongs to one level-1 bucket: its partitions are consecutive
    const uint64_t firstWord = CR ? static_cast<uint64_t>(crWord0(r.in, tile.begin)) : r.in[tile.begin * W];
    const uint64_t firstKey = (KR ? (twangMix64(firstWord) & r.slotMask) : firstWord) & ((1ULL << r.keyBits) - 1);
    const uint64_t bucket = (firstKey >> r.shiftB) >> r.shift2;
    const uint64_t part0 = bucket << r.shift2;
    const uint32_t cap = r.bucketCap[bucket];
    uint64_t w[R][W];
    auto loadSub = [&](uint32_t base) {   // (as in rpScatter2SortedBody)
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const uint32_t i = base + u * kSortThreads + threadIdx.x;
        recLoad<W, CR>(r.in, r.crCapIn, tile.begin + (i < tile.count ? i : tile.count - 1), w[u]);
      }
    };
    for (uint32_t base = 0; base < tile.count; base += SortLds<W>::kSub) {
      loadSub(base);
      uint32_t bin[R];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const uint32_t i = base + u * kSortThreads + threadIdx.x;
        const uint32_t part = static_cast<uint32_t>(KR ? (twangMix64(w[u][0]) & r.slotMask) : w[u][0]);
        bin[u] = i < tile.count ? ((part >> r.shiftB) & binMask) : 0xffffffffu;
      }
      rpSortedPlace<W, R>(l, r.numBins, w, bin, [&](uint32_t b, uint32_t count) -> unsigned long long {
        const uint32_t at = atomicAdd(&r.partCount[part0 + b], count);
        if (at + count > cap) {
          *r.overflow = 1;
          return ~0ULL;
        }
        return r.partBase[part0 + b] + at;
      });
      auto binOf = [&](uint64_t w0) {
        return (static_cast<uint32_t>(KR ? (twangMix64(w0) & r.slotMask) : w0) >> r.shiftB) & binMask;
      };
      rpSortedWrite<W, true, decltype(binOf)&, CR>(l, r.numBins, r.out, binOf, r.crCapOut);
    }
  }
}

struct RadixAggArgs {
  const uint64_t* recs;
  const uint64_t* partBegin;   // record offsets per histogram cell
  // optimistic level 2 (k_rp_scatter2_opt): partition p = partCount[p] records from partBase[p]
  const uint64_t* partBase;
  const uint32_t* partCount;
  const uint32_t* partCell;    // partition p -> cell (numParts + 1 entries); null: cell = p * cellStride
  int64_t cellStride;
  int64_t numParts;
  int32_t recWords;
  int32_t numAccs;
  int32_t shiftB;
  int32_t stride;
  uint64_t* table;
  uint64_t rowBase;
  uint64_t capacity;
  Counters* counters;
  int32_t kind[kRadixMaxAccs];
  int32_t off[kRadixMaxAccs];
  int32_t valIdx[kRadixMaxAccs];
  int32_t accOfVal[kRadixMaxAccs];
  uint64_t sliceRecs;          // records one workgroup folds at a time (see k_rp_aggregate)
  // LDS / table words of the fold: a DOUBLE sum owns two (hi on the grid, lo the
  // exact remainder: same split as every other aggregation kernel).
  int32_t numWords;
  int32_t ldsIdx[kRadixMaxAccs];            // first LDS word of accumulator j
  int32_t wordKind[2 * kRadixMaxAccs];
  int32_t wordOff[2 * kRadixMaxAccs];       // word offset inside the group row
  double splitM[kRadixMaxAccs];
  int32_t keyBits;                          // layout of record word 0
  int32_t rowBits;
  uint64_t crCap;                           // compact records: record capacity of 'recs' (0 = 16-byte records)
  uint64_t krSlotMask;                      // keyed records (hashRecLoad): home slot = twangMix64(key) & krSlotMask
  uint32_t krMask;                          // ... and every record's accumulator mask
  // virgin: the table has never been written (no k_init_table ran): the owner of a partition
  // stores every one of its group rows completely - untouched words from 'pattern' - instead of
  // read-modify-write, and empty partitions are initialised on the way.
  int32_t virgin;
  int32_t phase;                            // 0: the owners' launch, 1: the launch for the remaining slices of split partitions
  const uint64_t* pattern;                  // stride words of an empty group row
  int8_t ldsOfWord[2 + kMaxLdsAccs];        // word of the group row -> LDS word of the fold, -1 = none
  // (first input row, group row index) of every group this launch creates: finalize sorts these
  // instead of scanning the whole table for live rows (k_collect). A group's first row is final
  // when it is created - later chunks only bring larger row numbers - except in partitions split
  // into slices, which set counters->pairsBroken.
  uint64_t* pairKeys;
  uint32_t* pairVals;
  uint64_t pairBase;
  // dense (hashed folds into an operator without groups, see hashFoldFlushDense): 'table' is a plain
  // array of group rows that the folds APPEND to - row index = counters->numNewGroups before the
  // flush - instead of an open-addressing table; rows beyond denseCap are counted, not written;
  // denseFlags[0] = 1 when some key may own more than one row (split partitions, LDS overflow).
  int32_t dense;
  int32_t hashSlots;   // entries of the LDS table of a hashed fold: a power of two, 512 .. kHashSlots
  int32_t groupShift;  // log2(consecutive partitions folded into one LDS table and flushed together)
  uint64_t denseCap;
  // [0] = 1 when some key may own more than one row; [1] = rows handed out: a workgroup takes rows
  // in blocks of denseChunk (one atomic on this word per block, not per partition - a fold would
  // wait ~3 us for each) and leaves what it does not use of a block EMPTY (the pattern of an empty
  // group row, a pair that sorts behind every group): the array has holes, counters->numNewGroups
  // counts the groups
  uint32_t* denseFlags;
  uint32_t denseChunk;
  uint32_t pad2;
  // partitions folded in slices (more than sliceRecs records: skewed keys, few partitions): their
  // owners list them here - [0] = how many, entries from [16] - and the launch for the other slices
  // (phase 1) walks the list. (Every workgroup looking at every partition's range for them took
  // 3.3 of the 16.3 ms of the hashed fold of 2^20 partitions.)
  uint32_t* splitList;
};

// The accumulator plan of a fold, in registers: read from the kernel arguments ONCE with constant
// indices. (Indexing r.kind[j] / r.ldsIdx[j] with a run-time j made every record of every lane wait
// for three scalar loads from the argument buffer: the folds were bound by that, not by HBM.)
struct FoldPlan {
  int32_t numAccs;
  int32_t keyBits;
  int32_t rowBits;
  uint32_t rowMask;
  int32_t kind[kRadixMaxAccs];
  int32_t valIdx[kRadixMaxAccs];   // record word 1 + valIdx is the operand; < 0: the accumulator counts rows
  int32_t ldsIdx[kRadixMaxAccs];   // first LDS word of the accumulator
  double splitM[kRadixMaxAccs];
};

__device__ inline FoldPlan foldPlan(const RadixAggArgs& r) {
  FoldPlan fp;
  fp.numAccs = r.numAccs;
  fp.keyBits = r.keyBits;
  fp.rowBits = r.rowBits;
  fp.rowMask = static_cast<uint32_t>((1ULL << r.rowBits) - 1);
#pragma unroll
  for (int j = 0; j < kRadixMaxAccs; ++j) {
    fp.kind[j] = r.kind[j];
    fp.valIdx[j] = r.valIdx[j];
    fp.ldsIdx[j] = r.ldsIdx[j];
    fp.splitM[j] = r.splitM[j];
  }
  return fp;
}

// Word 'idx' (1 .. W - 1) of a record held in registers: a chain of selects, because a register
// array indexed with a run-time value is moved to scratch memory.
template <int W>
__device__ inline uint64_t recordWord(const uint64_t (&w)[W], int idx) {
  uint64_t v = 0;
#pragma unroll
  for (int q = 1; q < W; ++q) {
    v = q == idx ? w[q] : v;
  }
  return v;
}

// The operands of one record into the LDS accumulator words of its group ('acc' = the group's first word).
template <int W>
__device__ inline void foldAccumulate(const FoldPlan& fp, uint64_t* acc, const uint64_t (&w)[W], uint32_t mask,
                                      Counters* counters) {
#pragma unroll
  for (int j = 0; j < kRadixMaxAccs; ++j) {
    if (j >= fp.numAccs || !((mask >> j) & 1)) {
      continue;
    }
    uint64_t* word = acc + fp.ldsIdx[j];
    if (fp.valIdx[j] < 0) {
      applyLds(word, fp.kind[j], 1ULL, counters);
      continue;
    }
    const uint64_t v = recordWord<W>(w, 1 + fp.valIdx[j]);
    if (fp.kind[j] == ACC_SUM_F64 && fp.splitM[j] != 0.0) {
      double hi, lo;
      splitDouble(__longlong_as_double(static_cast<long long>(v)), fp.splitM[j], &hi, &lo);
      applyLds(word, ACC_SUM_F64, static_cast<uint64_t>(__double_as_longlong(hi)), counters);
      applyLds(word + 1, ACC_SUM_F64, static_cast<uint64_t>(__double_as_longlong(lo)), counters);
    } else {
      applyLds(word, fp.kind[j], v, counters);
    }
  }
}

// LDS state of one fold: acc[B][A] + first[B], A = LDS words per group.
struct RpFold {
  FoldPlan plan;
  uint64_t* acc;
  uint32_t* first;
  uint32_t* scratch;   // [0] groups created by this flush, [1] their base in the launch's pair list
  int B;
  int A;
};

__device__ inline void rpFoldInit(const RpFold& f, const RadixAggArgs& r) {
  for (int i = threadIdx.x; i < f.B * f.A; i += blockDim.x) {
    f.acc[i] = accIdentity(r.wordKind[i % f.A]);
  }
  for (int i = threadIdx.x; i < f.B; i += blockDim.x) {
    f.first[i] = 0xffffffffu;
  }
  blockSync();
}

// One record applied to the LDS accumulators of its group.
template <int W>
__device__ inline void rpFoldRecord(const RpFold& f, const RadixAggArgs& r, const uint64_t (&w)[W]) {
  const FoldPlan& fp = f.plan;
  const uint64_t w0 = w[0];
  const uint32_t g = static_cast<uint32_t>(w0) & static_cast<uint32_t>(f.B - 1);
  const uint32_t row = static_cast<uint32_t>(w0 >> fp.keyBits) & fp.rowMask;
  const uint32_t mask = static_cast<uint32_t>(w0 >> (fp.keyBits + fp.rowBits));
  if (f.first[g] > row) {
    atomicMin(&f.first[g], row);
  }
  foldAccumulate<W>(fp, f.acc + static_cast<size_t>(g) * f.A, w, mask, r.counters);
}

// Folds records [begin, end) of one partition into the LDS accumulators.
template <int W, bool CR = false>
__device__ inline void rpFoldRecords(const RpFold& f, const RadixAggArgs& r, uint64_t begin, uint64_t end) {
  // kRadixUnroll / 2 records per lane and round, the next round's loaded before this round's go through the
  // LDS atomics: the same records in flight as eight per round, but at every moment (round 6: the dense fold
  // of config 4 4.3 - 4.4 -> 4.0 ms, tools/r06_fold.sh)
  constexpr int kHalf = kRadixUnroll / 2;
  if (begin >= end) {  // (uniform; the clamped loads below need one record)
    blockSync();
    return;
  }
  uint64_t nxt[kHalf][W];
#pragma unroll
  for (int u = 0; u < kHalf; ++u) {
    const uint64_t i = begin + u * 512 + threadIdx.x;
    recLoad<W, CR>(r.recs, r.crCap, i < end ? i : end - 1, nxt[u]);
  }
  for (uint64_t at = begin; at < end; at += kHalf * 512) {
    uint64_t w[kHalf][W];
#pragma unroll
    for (int u = 0; u < kHalf; ++u) {
#pragma unroll
      for (int q = 0; q < W; ++q) {
        w[u][q] = nxt[u][q];
      }
    }
    if (at + kHalf * 512 < end) {
#pragma unroll
      for (int u = 0; u < kHalf; ++u) {
        const uint64_t i = at + kHalf * 512 + u * 512 + threadIdx.x;
        recLoad<W, CR>(r.recs, r.crCap, i < end ? i : end - 1, nxt[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < kHalf; ++u) {
      const uint64_t i = at + u * 512 + threadIdx.x;
      if (i < end) {
        rpFoldRecord<W>(f, r, w[u]);
      }
    }
  }
  blockSync();
}

// A complete group row stored: pairs of words as 16-byte stores when the stride is even (rows are
// then 16-byte aligned: four 8-byte stores per 32-byte row made the fold write 8.7 GB for 5.5).
template <typename WordFn>
__device__ inline void storeRowWords(uint64_t* row, int stride, WordFn&& word) {
  if ((stride & 1) == 0) {
    for (int w = 0; w < stride; w += 2) {
      RpU64x2 v;
      v.x = word(w);
      v.y = word(w + 1);
      *reinterpret_cast<RpU64x2*>(row + w) = v;
    }
  } else {
    for (int w = 0; w < stride; ++w) {
      row[w] = word(w);
    }
  }
}

// One group of a fold added into its group row (see rpFoldFlush); true = the group is new.
__device__ inline bool rpFlushGroup(const RpFold& f, const RadixAggArgs& r, int g, uint64_t base, bool virgin,
                                    bool exclusive, bool hasRecords) {
  const int A = f.A;
  const uint32_t fr = hasRecords ? f.first[g] : 0xffffffffu;
  uint64_t* row = r.table + (base + g) * r.stride;
  const uint64_t mine = r.rowBase + static_cast<uint64_t>(fr);
  bool isNew = false;
  if (virgin) {
    isNew = fr != 0xffffffffu;
    storeRowWords(row, r.stride, [&](int w) {
      uint64_t v = r.pattern[w];
      if (isNew) {
        const int j = r.ldsOfWord[w];
        v = w == 1 ? mine : (j >= 0 ? f.acc[static_cast<size_t>(g) * A + j] : v);
      }
      return v;
    });
  } else if (fr == 0xffffffffu) {
    // nothing for this group
  } else if (!exclusive) {
    const unsigned long long old = atomicMin(reinterpret_cast<unsigned long long*>(row + 1), mine);
    isNew = old == kNoRow;
    for (int j = 0; j < A; ++j) {
      const uint64_t v = f.acc[static_cast<size_t>(g) * A + j];
      const int32_t kind = r.wordKind[j];
      if (v != accIdentity(kind)) {
        if (kind == ACC_SUM_I64) {
          addPartial128Global(row + r.wordOff[j], v, 0);  // its high word follows as word j + 1
        } else {
          applyGlobal(row + r.wordOff[j], kind == ACC_COUNT ? ACC_SUM_I64_WRAP : kind, v, r.counters);
        }
      }
    }
  } else {
    const uint64_t old = row[1];
    isNew = old == kNoRow;
    if (mine < old) {
      row[1] = mine;
    }
    for (int j = 0; j < A; ++j) {
      const uint64_t v = f.acc[static_cast<size_t>(g) * A + j];
      uint64_t* word = row + r.wordOff[j];
      switch (r.wordKind[j]) {
        case ACC_SUM_F64:
          *reinterpret_cast<double*>(word) += __longlong_as_double(static_cast<long long>(v));
          break;
        case ACC_SUM_I64: {
          const uint64_t before = *word;
          *word = before + v;
          word[1] += static_cast<uint64_t>(carryUnsigned(before, v));  // the fold's own high word is word j + 1
          break;
        }
        case ACC_SUM_I64_HI:
        case ACC_SUM_I64_WRAP:
        case ACC_COUNT:
          *word += v;
          break;
        case ACC_MIN:
          *word = v < *word ? v : *word;
          break;
        default:
          *word = v > *word ? v : *word;
          break;
      }
    }
  }
  return isNew;
}

// Adds the LDS block into the partition's group rows. exclusive: this workgroup is the only
// writer of those rows during the launch — plain read-modify-write, coalesced over consecutive
// groups (virgin table: plain stores of complete rows, nothing is read); otherwise (the partition
// is shared by several workgroups, see below) HBM atomics. hasRecords = false: an empty partition
// of a virgin table, whose rows only get their initial pattern. Groups created here are counted
// with ONE atomic per flush and listed as (first row, group row) pairs.
constexpr int kRpMaxPerThread = 8;  // B <= 4096 groups, 512 threads

__device__ inline void rpFoldFlush(const RpFold& f, const RadixAggArgs& r, int64_t p, bool exclusive, bool hasRecords) {
  const uint64_t base = static_cast<uint64_t>(p) << r.shiftB;
  const bool virgin = r.virgin != 0 && exclusive;
  if (threadIdx.x == 0) {
    f.scratch[0] = 0;
    if (!exclusive) {
      r.counters->pairsBroken = 1;
    }
  }
  blockSync();
  uint32_t myPos[kRpMaxPerThread];
  int k = 0;
  for (int g = threadIdx.x; g < f.B; g += blockDim.x, ++k) {
    myPos[k] = 0xffffffffu;
    if (base + g >= r.capacity) {
      continue;
    }
    const bool isNew = rpFlushGroup(f, r, g, base, virgin, exclusive, hasRecords);
    if (isNew) {
      myPos[k] = atomicAdd(&f.scratch[0], 1u);
    }
  }
  blockSync();
  if (threadIdx.x == 0 && f.scratch[0] != 0) {
    f.scratch[1] = atomicAdd(&r.counters->numNewGroups, f.scratch[0]);
  }
  blockSync();
  if (r.pairKeys != nullptr && f.scratch[0] != 0) {
    k = 0;
    for (int g = threadIdx.x; g < f.B; g += blockDim.x, ++k) {
      if (myPos[k] != 0xffffffffu) {
        const uint64_t at = r.pairBase + f.scratch[1] + myPos[k];
        r.pairKeys[at] = r.rowBase + static_cast<uint64_t>(f.first[g]);
        r.pairVals[at] = static_cast<uint32_t>(base + g);
      }
    }
  }
  if (hasRecords) {
    // the groups go back to "nothing seen": the block is initialised once per launch, not per partition
    for (int g = threadIdx.x; g < f.B; g += blockDim.x) {
      if (f.first[g] != 0xffffffffu) {
        f.first[g] = 0xffffffffu;
        for (int j = 0; j < f.A; ++j) {
          f.acc[static_cast<size_t>(g) * f.A + j] = accIdentity(r.wordKind[j]);
        }
      }
    }
  }
  blockSync();  // the LDS block is reused by the next partition
}

__device__ inline void rpPartitionRange(const RadixAggArgs& r, int64_t p, uint64_t* begin, uint64_t* end) {
  if (r.partBase != nullptr) {
    *begin = r.partBase[p];
    *end = *begin + r.partCount[p];
    return;
  }
  *begin = r.partBegin[r.partCell ? r.partCell[p] : p * r.cellStride];
  *end = r.partBegin[r.partCell ? r.partCell[p + 1] : (p + 1) * r.cellStride];
}

// Fold: one workgroup per partition folds its records into LDS, then touches
// each of the <= B group rows in HBM once. A partition with more than
// sliceRecs records (few partitions, or skewed keys) is cut into slices:
// its owner folds the first one, the others are spread over all workgroups
// in a second phase, and every slice of such a partition is flushed with
// atomics instead of plain read-modify-write.
// (Measured and dropped, round 5: the next partition's first chunk of records loading while the current one is
// flushed - 122 VGPRs with the chunk live across the flush, two workgroups per CU instead of three: 5.7 ms
// instead of 4.2 for config 4's 12-byte records, 5.2 either way for 16-byte ones. With two register buffers
// and loads a chunk ahead inside the partition as well: 131 VGPRs, one workgroup per CU.)
template <int W, bool CR = false>
__global__ __launch_bounds__(512) void k_rp_aggregate(RadixAggArgs r) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
  __shared__ uint32_t scratch[2];
  RpFold f;
  f.plan = foldPlan(r);
  f.B = 1 << r.shiftB;
  f.A = r.numWords;
  f.acc = reinterpret_cast<uint64_t*>(ldsRaw);                                       // [B][A]
  f.first = reinterpret_cast<uint32_t*>(f.acc + static_cast<size_t>(f.B) * f.A);   // [B]
  f.scratch = scratch;
  rpFoldInit(f, r);   // once: every flush leaves the block as it found it
  if (r.phase != 1) {
    for (int64_t p = blockIdx.x; p < r.numParts; p += gridDim.x) {
      uint64_t begin, end;
      rpPartitionRange(r, p, &begin, &end);
      if (end == begin) {  // uniform per workgroup
        if (r.virgin) {
          rpFoldFlush(f, r, p, true, false);  // its rows still need their initial pattern
        }
        continue;
      }
      const bool split = end - begin > r.sliceRecs;
      rpFoldRecords<W, CR>(f, r, begin, split ? begin + r.sliceRecs : end);
      // the owner of a split partition of a virgin table stores complete rows like any owner: the
      // other slices run in the next launch (phase 1), behind the launch boundary
      rpFoldFlush(f, r, p, !split || r.virgin != 0, true);
      if (split && threadIdx.x == 0) {
        r.counters->pairsBroken = 1;
        r.splitList[16 + atomicAdd(&r.splitList[0], 1u)] = static_cast<uint32_t>(p);
      }
    }
  }
  if (r.phase == 0) {
    return;
  }
  // Remaining slices of the split partitions (a launch of its own: r.splitList is complete).
  const uint32_t numSplit = r.splitList[0];
  for (uint32_t k = 0; k < numSplit; ++k) {
    const int64_t p = r.splitList[16 + k];
    uint64_t begin, end;
    rpPartitionRange(r, p, &begin, &end);
    const uint64_t slices = (end - begin + r.sliceRecs - 1) / r.sliceRecs;
    for (uint64_t s = 1 + blockIdx.x; s < slices; s += gridDim.x) {
      const uint64_t b = begin + s * r.sliceRecs;
      rpFoldRecords<W, CR>(f, r, b, b + r.sliceRecs < end ? b + r.sliceRecs : end);
      rpFoldFlush(f, r, p, false, true);
    }
  }
}

// ---- fold for open-addressing tables (normalized-key mode; BASELINE config 4 with sparse keys) ----
// k_agg_global pays two or three HBM atomics per row (one per accumulator word + the first-row
// word), and the chip retires ~21 G of them per second: 10^9 rows take > 100 ms whatever else
// happens. Here the rows arrive partitioned by the HOME SLOT of their group in the global table
// (twang_mix64(key) & slotMask), in partitions of ~1024 records: the table grows with the number
// of groups, a chunk of rows does not, so the slot range of a partition is chosen per launch
// (capacity / partitions), not fixed. The workgroup that owns a partition folds its records into an
// LDS hash table of kHashSlots entries {key, accumulator words, first row} - start position = the
// home slot scaled into the table (entries stay in home-slot order), linear probing, entries
// claimed with LDS compare-and-swap - and then visits every occupied entry's group row in HBM
// ONCE: findOrInsert from the home slot (ascending over the lanes: the rows of a partition are
// neighbours in the table), plain read-modify-write, because a key has exactly one home slot and
// therefore exactly one owner per launch. A record that finds the LDS table full goes to its group
// row with HBM atomics on its own (and makes the whole partition flush with atomics).
// The table has r.hashSlots entries: kHashSlots when nothing is known (load <= 0.5 even if all the
// records of a partition are distinct), fewer - down to 512 - when a sample of the partitions
// (k_rp_distinct_sample, behind the level-2 scatter) says that their keys repeat: initialising and
// scanning 2048 entries for ~100 keys was a quarter of the fold (config 4, sparse keys: 21.8 ->
// 16.1 ms together with the loads running one partition ahead, hashFoldLoadAhead).
// What did NOT move the fold, each measured on that workload: one atomic per block of rows instead
// of one per partition for the dense flush (-1 ms; kept), the accumulator plan in registers instead
// of scalar loads from the argument buffer per record (0), three workgroups per CU at 80 registers
// with spills instead of two (0), consecutive instead of strided partitions per workgroup (0; the
// direct-index fold lost 2 ms), one WAVE per partition without any barrier for the direct-index
// fold (0). With the fold and the flush switched off the loop still takes 7 ms for 24 GB.
constexpr int kHashSlots = 2048;           // most LDS entries per fold: 56 KB with two words, two workgroups per CU
constexpr int kHashRecsPerPart = 1024;     // records a partition is sized for (load <= 0.5 if all distinct)

struct HashFold {
  FoldPlan plan;
  int posUp, posDown;        // home slot inside the partition -> entry of the LDS table: (x << posUp) >> posDown
  unsigned long long* keys;  // [S], kEmpty = free
  uint64_t* acc;             // [S][A]
  uint32_t* first;           // [S]
  uint32_t* scratch;         // [0] new groups, [1] pair base, [2] some record overflowed the window
  uint32_t* alloc;           // dense folds: the workgroup's block of rows, see hashFoldFlushDense
  int S;
  int A;
};

__device__ inline void hashFoldInit(const HashFold& f, const RadixAggArgs& r) {
  for (int i = threadIdx.x; i < f.S; i += blockDim.x) {
    f.keys[i] = kEmpty;
    f.first[i] = 0xffffffffu;
  }
  for (int i = threadIdx.x; i < f.S * f.A; i += blockDim.x) {
    f.acc[i] = accIdentity(r.wordKind[i % f.A]);
  }
  if (threadIdx.x == 0) {
    f.scratch[2] = 0;
  }
  blockSync();
}

// An entry a flush has read goes back to "free": the table is initialised once per launch, not
// once per partition (a pass over all entries and a barrier less per fold).
__device__ inline void hashFoldResetEntry(const HashFold& f, const RadixAggArgs& r, int e) {
  f.keys[e] = kEmpty;
  f.first[e] = 0xffffffffu;
  for (int j = 0; j < f.A; ++j) {
    f.acc[static_cast<size_t>(e) * f.A + j] = accIdentity(r.wordKind[j]);
  }
}

// The operands of one record applied to a group row in HBM with atomics: what updateGlobal does.
template <int W>
__device__ inline void hashApplyRecordGlobal(const RadixAggArgs& r, uint64_t* g, const uint64_t (&w)[W], uint32_t mask) {
  for (int j = 0; j < r.numAccs; ++j) {
    if (!((mask >> j) & 1)) {
      continue;
    }
    const uint64_t v = r.valIdx[j] < 0 ? 1ULL : recordWord<W>(w, 1 + r.valIdx[j]);
    uint64_t* word = g + r.wordOff[r.ldsIdx[j]];
    if (r.kind[j] == ACC_SUM_F64 && r.splitM[j] != 0.0) {
      double hi, lo;
      splitDouble(__longlong_as_double(static_cast<long long>(v)), r.splitM[j], &hi, &lo);
      applyGlobal(word, ACC_SUM_F64, static_cast<uint64_t>(__double_as_longlong(hi)), r.counters);
      applyGlobal(word + 1, ACC_SUM_F64, static_cast<uint64_t>(__double_as_longlong(lo)), r.counters);
    } else {
      applyGlobal(word, r.kind[j] == ACC_COUNT ? ACC_SUM_I64_WRAP : r.kind[j], v, r.counters);
    }
  }
}

// One record straight to its group row in HBM (window full).
template <int W, bool DENSE>
__device__ inline void hashFoldDirect(const RadixAggArgs& r, const uint64_t (&w)[W], uint64_t key, uint32_t row,
                                      uint32_t mask) {
  if constexpr (DENSE) {
    // a row of its own; the merge pass (k_dense_merge) brings the rows of one key together
    const uint64_t idx = atomicAdd(&r.denseFlags[1], 1u);
    atomicAdd(&r.counters->numNewGroups, 1u);
    r.denseFlags[0] = 1;
    if (idx >= r.denseCap) {
      return;
    }
    uint64_t* g = r.table + idx * r.stride;
    for (int x = 0; x < r.stride; ++x) {
      g[x] = r.pattern[x];
    }
    g[0] = key;
    g[1] = r.rowBase + row;
    hashApplyRecordGlobal<W>(r, g, w, mask);
    return;
  }
  uint64_t* g = findOrInsert(r.table, r.stride, r.capacity, key, r.counters);
  if (g == nullptr) {
    return;
  }
  const unsigned long long old = atomicMin(reinterpret_cast<unsigned long long*>(g + 1), r.rowBase + row);
  if (old == kNoRow) {
    atomicAdd(&r.counters->numNewGroups, 1u);
    r.counters->pairsBroken = 1;  // this group is not in the launch's pair list
  }
  hashApplyRecordGlobal<W>(r, g, w, mask);
}

// Keyed records (KR; round 5): when nobody asked for first-seen order and every record has the same accumulator
// mask (flat operands without nulls or masks), word 0 of a hashed record says nothing the key does not: the home
// slot is twangMix64(key) & slotMask - a dozen ALU operations against 8 bytes through three passes. The record
// in HBM is {key, operand}, 16 bytes instead of 24 (and sub-tiles of 8192 instead of 4096 records in the
// scatters' LDS); the folds load it into the three-word form with word 0 unused.
template <int W, bool KR>
__device__ inline void hashRecLoad(const RadixAggArgs& r, uint64_t i, uint64_t (&w)[W]) {
  if constexpr (KR) {
    static_assert(W == 3, "keyed records carry one operand");
    uint64_t t[2];
    rpLoad<2>(r.recs + i * 2, t);
    w[0] = 0;
    w[1] = t[1];
    w[2] = t[0];
  } else {
    rpLoad<W>(r.recs + i * W, w);
  }
}

// One record folded into the LDS table of its partition.
template <int W, bool DENSE, bool KR = false>
__device__ inline void hashFoldRecord(const HashFold& f, const RadixAggArgs& r, uint64_t base, const uint64_t (&w)[W]) {
  const FoldPlan& fp = f.plan;
  const uint64_t w0 = w[0];
  const uint64_t key = w[W - 1];
  const uint32_t row = KR ? 0u : static_cast<uint32_t>(w0 >> fp.keyBits) & fp.rowMask;
  const uint32_t mask = KR ? r.krMask : static_cast<uint32_t>(w0 >> (fp.keyBits + fp.rowBits));
  const uint64_t home = KR ? (twangMix64(key) & r.krSlotMask) : (w0 & ((1ULL << fp.keyBits) - 1));
  // home slot inside the partition, scaled into the LDS table (both sizes are powers of two)
  int pos = static_cast<int>(((home - base) << f.posUp) >> f.posDown);
  uint32_t seenFirst = 0xffffffffu;
  for (int probes = 0;; ++probes) {
    const unsigned long long k = f.keys[pos];
    // read with the key, not behind it (one LDS round trip less per record); only a filter for the
    // atomic below: a stale value is too high, never too low
    seenFirst = f.first[pos];
    if (k == key) {
      break;
    }
    if (k == kEmpty) {
      const unsigned long long old = atomicCAS(&f.keys[pos], kEmpty, static_cast<unsigned long long>(key));
      if (old == kEmpty || old == key) {
        break;
      }
    }
    if (probes >= f.S) {
      pos = -1;  // table full
      break;
    }
    pos = (pos + 1) & (f.S - 1);
  }
  if (pos < 0) {
    f.scratch[2] = 1;
    hashFoldDirect<W, DENSE>(r, w, key, row, mask);
    return;
  }
  if (seenFirst > row) {
    atomicMin(&f.first[pos], row);
  }
  foldAccumulate<W>(fp, f.acc + static_cast<size_t>(pos) * f.A, w, mask, r.counters);
}

constexpr int kHashUnroll = 4;  // records per lane in flight beyond the ones loaded ahead

template <int W, bool DENSE, bool KR = false>
__device__ inline void hashFoldRecords(const HashFold& f, const RadixAggArgs& r, uint64_t base, uint64_t begin, uint64_t end) {
  for (uint64_t at = begin; at < end; at += kHashUnroll * 512) {
    uint64_t w[kHashUnroll][W];
#pragma unroll
    for (int u = 0; u < kHashUnroll; ++u) {
      const uint64_t i = at + u * 512 + threadIdx.x;
      hashRecLoad<W, KR>(r, i < end ? i : end - 1, w[u]);  // clamped, unconditional: see k_rp_scatter2
    }
#pragma unroll
    for (int u = 0; u < kHashUnroll; ++u) {
      const uint64_t i = at + u * 512 + threadIdx.x;
      if (i < end) {
        hashFoldRecord<W, DENSE, KR>(f, r, base, w[u]);
      }
    }
  }
  blockSync();
}

// The first kHashAhead x 512 records of a partition travel through registers one partition ahead
// of the fold: a fold is a chain of HBM round trips (range, records, the group counter), and the
// two or three workgroups a CU holds do not cover it (measured: 11 us per partition, 1.1 TB/s).
constexpr int kHashAhead = 2;

template <int W, bool KR = false>
__device__ inline void hashFoldLoadAhead(const RadixAggArgs& r, uint64_t begin, uint64_t end, uint64_t (&w)[kHashAhead][W]) {
  const uint64_t stop = end - begin > r.sliceRecs ? begin + r.sliceRecs : end;
#pragma unroll
  for (int u = 0; u < kHashAhead; ++u) {
    const uint64_t i = begin + u * 512 + threadIdx.x;
    if (begin < stop) {  // uniform
      hashRecLoad<W, KR>(r, i < stop ? i : stop - 1, w[u]);  // clamped, unconditional: see k_rp_scatter2
    }
  }
}

// Range of partition p through the vector memory path (a buffer load: the index is uniform, and a
// scalar load would be waited for at the next barrier together with the LDS).
__device__ inline void rpPartitionRangeAhead(const RadixAggArgs& r, int64_t p, uint64_t* begin, uint64_t* end) {
  if (r.partBase == nullptr) {
    rpPartitionRange(r, p, begin, end);
    return;
  }
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const auto bases = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(r.partBase), 0,
                                                      static_cast<int>(r.numParts * 8), 0x00020000);
  const auto counts = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(r.partCount), 0,
                                                       static_cast<int>(r.numParts * 4), 0x00020000);
  const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(bases, static_cast<int>(p * 8), 0, 0);
  const uint32_t c = __builtin_amdgcn_raw_buffer_load_b32(counts, static_cast<int>(p * 4), 0, 0);
  *begin = static_cast<uint64_t>(b.x) | (static_cast<uint64_t>(b.y) << 32);
  *end = *begin + c;
}

// Flush: every lane owns kHashPerLane entries of the LDS table and works on all of them at once -
// the probe loads, the claims and the row words of its entries are issued back to back, so a lane
// keeps several dependent HBM round trips in flight instead of one (the flush is latency bound:
// one findOrInsert + one read-modify-write per group).
constexpr int kHashPerLane = kHashSlots / 512;

__device__ inline void hashFoldFlush(const HashFold& f, const RadixAggArgs& r, bool exclusive) {
  const int A = f.A;
  exclusive = exclusive && f.scratch[2] == 0;  // direct records touched rows of this partition with atomics
  if (threadIdx.x == 0) {
    f.scratch[0] = 0;
    if (!exclusive) {
      r.counters->pairsBroken = 1;
    }
  }
  blockSync();
  const uint64_t mask = r.capacity - 1;
  uint64_t key[kHashPerLane];
  uint64_t pos[kHashPerLane];
  uint64_t* row[kHashPerLane];
  bool pending[kHashPerLane];
  uint32_t myPos[kHashPerLane];
#pragma unroll
  for (int k = 0; k < kHashPerLane; ++k) {
    // consecutive lanes take consecutive entries: the table is in home-slot order, so a wave's
    // probes walk the global table in ascending order
    const int e = k * 512 + threadIdx.x;
    key[k] = e < f.S ? f.keys[e] : kEmpty;
    pending[k] = key[k] != kEmpty;
    pos[k] = twangMix64(key[k]) & mask;
    row[k] = nullptr;
    myPos[k] = 0xffffffffu;
  }
  // findOrInsert for all entries of the lane together: one round = one probe of each pending entry
  for (uint64_t round = 0; round <= mask; ++round) {
    uint64_t seen[kHashPerLane];
    bool any = false;
#pragma unroll
    for (int k = 0; k < kHashPerLane; ++k) {
      seen[k] = pending[k] ? __hip_atomic_load(r.table + pos[k] * r.stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                           : 0;
    }
#pragma unroll
    for (int k = 0; k < kHashPerLane; ++k) {
      if (!pending[k]) {
        continue;
      }
      uint64_t* g = r.table + pos[k] * r.stride;
      uint64_t found = seen[k];
      if (found == kEmpty) {
        found = atomicCAS(reinterpret_cast<unsigned long long*>(g), kEmpty, static_cast<unsigned long long>(key[k]));
        if (found == kEmpty) {
          found = key[k];
        }
      }
      if (found == key[k]) {
        row[k] = g;
        pending[k] = false;
      } else {
        pos[k] = (pos[k] + 1) & mask;
        any = true;
      }
    }
    if (!any) {
      break;
    }
    if (round == mask) {
      r.counters->tableFull = 1;
    }
  }
  // the rows' first-row words, all at once
  uint64_t oldFirst[kHashPerLane];
#pragma unroll
  for (int k = 0; k < kHashPerLane; ++k) {
    oldFirst[k] = (row[k] != nullptr && exclusive) ? row[k][1] : 0;
  }
#pragma unroll
  for (int k = 0; k < kHashPerLane; ++k) {
    if (row[k] == nullptr) {
      continue;
    }
    const int e = k * 512 + threadIdx.x;
    uint64_t* g = row[k];
    const uint64_t mine = r.rowBase + static_cast<uint64_t>(f.first[e]);
    bool isNew;
    if (!exclusive) {
      const unsigned long long old = atomicMin(reinterpret_cast<unsigned long long*>(g + 1), mine);
      isNew = old == kNoRow;
      for (int j = 0; j < A; ++j) {
        const uint64_t v = f.acc[static_cast<size_t>(e) * A + j];
        const int32_t kind = r.wordKind[j];
        if (v != accIdentity(kind)) {
          if (kind == ACC_SUM_I64) {
            addPartial128Global(g + r.wordOff[j], v, 0);
          } else {
            applyGlobal(g + r.wordOff[j], kind == ACC_COUNT ? ACC_SUM_I64_WRAP : kind, v, r.counters);
          }
        }
      }
    } else {
      isNew = oldFirst[k] == kNoRow;
      if (mine < oldFirst[k]) {
        g[1] = mine;
      }
      for (int j = 0; j < A; ++j) {
        const uint64_t v = f.acc[static_cast<size_t>(e) * A + j];
        uint64_t* word = g + r.wordOff[j];
        switch (r.wordKind[j]) {
          case ACC_SUM_F64:
            // a fresh group row holds the identity: no read needed
            *reinterpret_cast<double*>(word) =
                (isNew ? 0.0 : *reinterpret_cast<double*>(word)) + __longlong_as_double(static_cast<long long>(v));
            break;
          case ACC_SUM_I64: {
            const uint64_t before = isNew ? 0 : *word;
            *word = before + v;
            const uint64_t up = static_cast<uint64_t>(carryUnsigned(before, v));
            if (up != 0) {
              word[1] += up;
            }
            break;
          }
          case ACC_SUM_I64_HI:
          case ACC_SUM_I64_WRAP:
          case ACC_COUNT:
            *word = (isNew ? 0 : *word) + v;
            break;
          case ACC_MIN:
            *word = (isNew || v < *word) ? v : *word;
            break;
          default:
            *word = (isNew || v > *word) ? v : *word;
            break;
        }
      }
    }
    if (isNew) {
      myPos[k] = atomicAdd(&f.scratch[0], 1u);
    }
  }
  blockSync();
  if (threadIdx.x == 0 && f.scratch[0] != 0) {
    f.scratch[1] = atomicAdd(&r.counters->numNewGroups, f.scratch[0]);
  }
  blockSync();
  if (r.pairKeys != nullptr && f.scratch[0] != 0) {
#pragma unroll
    for (int k = 0; k < kHashPerLane; ++k) {
      if (myPos[k] != 0xffffffffu) {
        const int e = k * 512 + threadIdx.x;
        const uint64_t at = r.pairBase + f.scratch[1] + myPos[k];
        r.pairKeys[at] = r.rowBase + static_cast<uint64_t>(f.first[e]);
        r.pairVals[at] = static_cast<uint32_t>((row[k] - r.table) / r.stride);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kHashPerLane; ++k) {
    if (key[k] != kEmpty) {
      hashFoldResetEntry(f, r, k * 512 + threadIdx.x);
    }
  }
  if (threadIdx.x == 0) {
    f.scratch[2] = 0;
  }
  blockSync();
}

// Flush of a fold into an operator that has no groups yet: nothing to look up, so the fold's
// entries are APPENDED to a plain array of group rows, complete rows stored side by side
// (coalesced) instead of one findOrInsert + one read-modify-write per group at random places of an
// open-addressing table. A workgroup owns a block of rows at a time (f.alloc: {cursor, limit},
// double buffered so that the lanes read one copy while lane 0 writes the other) and takes the
// next one with ONE atomic on denseFlags[1]; its entries are counted with ballots. Rows of one key
// come from one fold only (a key has one partition), except when a partition is folded in slices
// or a record overflowed the LDS table: denseFlags[0] tells the host to merge (k_dense_merge).
enum { DA_CURSOR = 0, DA_LIMIT = 2, DA_WAVES = 4, DA_START = 12, DA_WORDS = 16 };

// Rows [from, to) of the dense array stay empty.
__device__ inline void denseFillHoles(const RadixAggArgs& r, uint32_t from, uint32_t to) {
  for (uint64_t i = static_cast<uint64_t>(from) + threadIdx.x; i < to && i < r.denseCap; i += blockDim.x) {
    uint64_t* g = r.table + i * r.stride;
    storeRowWords(g, r.stride, [&](int x) { return r.pattern[x]; });
    if (r.pairKeys != nullptr) {
      r.pairKeys[r.pairBase + i] = ~0ULL;
      r.pairVals[r.pairBase + i] = 0;
    }
  }
}

__device__ inline void hashFoldFlushDense(const HashFold& f, const RadixAggArgs& r, bool owner, int* parity,
                                          uint32_t* newGroups) {
  const int A = f.A;
  if (threadIdx.x == 0 && !owner) {
    r.denseFlags[0] = 1;
  }
  const int wave = threadIdx.x >> 6;
  uint32_t rank[kHashPerLane];
  uint32_t waveTotal = 0;
#pragma unroll
  for (int k = 0; k < kHashPerLane; ++k) {
    const int e = k * 512 + threadIdx.x;
    const bool live = e < f.S && f.keys[e] != kEmpty;
    const uint64_t m = ballot(live);
    rank[k] = live ? waveTotal + static_cast<uint32_t>(lanePrefix(m)) : 0xffffffffu;
    waveTotal += static_cast<uint32_t>(popc64(m));
  }
  if (lane() == 0) {
    f.alloc[DA_WAVES + wave] = waveTotal;
  }
  blockSync();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const uint32_t t = f.alloc[DA_WAVES + w];
    before += w < wave ? t : 0;
    total += t;
  }
  const int now = *parity;
  const uint32_t cursor = f.alloc[DA_CURSOR + now];
  const uint32_t limit = f.alloc[DA_LIMIT + now];
  uint32_t base = cursor;
  if (cursor + total <= limit) {  // uniform
    if (threadIdx.x == 0) {
      f.alloc[DA_CURSOR + (now ^ 1)] = cursor + total;
      f.alloc[DA_LIMIT + (now ^ 1)] = limit;
    }
  } else {
    // the next block; what is left of this one stays empty
    if (threadIdx.x == 0) {
      const uint32_t size = total > r.denseChunk ? total : r.denseChunk;
      const uint32_t start = atomicAdd(&r.denseFlags[1], size);
      f.alloc[DA_START] = start;
      f.alloc[DA_CURSOR + (now ^ 1)] = start + total;
      f.alloc[DA_LIMIT + (now ^ 1)] = start + size;
    }
    blockSync();
    base = f.alloc[DA_START];
    denseFillHoles(r, cursor, limit);
  }
  *parity = now ^ 1;
  *newGroups += total;
#pragma unroll
  for (int k = 0; k < kHashPerLane; ++k) {
    if (rank[k] == 0xffffffffu) {
      continue;
    }
    const int e = k * 512 + threadIdx.x;
    const uint64_t idx = static_cast<uint64_t>(base) + before + rank[k];
    if (idx < r.denseCap) {   // (beyond: counted - the host grows the array and folds again)
      uint64_t* g = r.table + idx * r.stride;
      const uint64_t first = r.rowBase + static_cast<uint64_t>(f.first[e]);
      storeRowWords(g, r.stride, [&](int x) {
        const int j = r.ldsOfWord[x];
        return x == 0 ? static_cast<uint64_t>(f.keys[e])
                      : (x == 1 ? first : (j >= 0 ? f.acc[static_cast<size_t>(e) * A + j] : r.pattern[x]));
      });
      if (r.pairKeys != nullptr) {
        r.pairKeys[r.pairBase + idx] = first;
        r.pairVals[r.pairBase + idx] = static_cast<uint32_t>(idx);
      }
    }
    hashFoldResetEntry(f, r, e);
  }
  if (threadIdx.x == 0) {
    f.scratch[2] = 0;
  }
  blockSync();
}

template <int W, bool DENSE, bool KR = false>
__global__ __launch_bounds__(512) void k_rp_aggregate_hashed(RadixAggArgs r) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
  __shared__ uint32_t scratch[4];
  __shared__ uint32_t alloc[DA_WORDS];
  HashFold f;
  f.alloc = alloc;
  int parity = 0;
  uint32_t newGroups = 0;
  if (threadIdx.x < DA_WORDS) {
    alloc[threadIdx.x] = 0;  // (hashFoldInit below ends with a barrier)
  }
  f.plan = foldPlan(r);
  f.S = r.hashSlots;
  // Partitions per flush (round 5): 2^groupShift consecutive partitions - a contiguous range of home slots - are
  // folded into ONE LDS table and flushed together when a sample of the keys said that they are few
  // (launchRadix): a partition is ~1000 records, and a flush per partition (barriers, the block of rows, the
  // groups' stores) cost more than its records.
  const int gs = r.groupShift;
  const int64_t G = 1LL << gs;
  {
    const int slotBits = 31 - __builtin_clz(static_cast<unsigned>(r.hashSlots));
    const int span = r.shiftB + gs;   // log2(home slots of one table)
    f.posUp = slotBits > span ? slotBits - span : 0;
    f.posDown = span > slotBits ? span - slotBits : 0;
  }
  f.A = r.numWords;
  f.keys = reinterpret_cast<unsigned long long*>(ldsRaw);
  f.acc = reinterpret_cast<uint64_t*>(f.keys + f.S);
  f.first = reinterpret_cast<uint32_t*>(f.acc + static_cast<size_t>(f.S) * f.A);
  f.scratch = scratch;
  hashFoldInit(f, r);   // once: every flush leaves the table empty again
  // software pipeline over the workgroup's partitions: the first records of the next DEPTH partitions
  // are in flight, their ranges one partition further (DEPTH = 2 at 128 registers: the same 16.4 ms
  // for 10^9 three-word records - what bounds the loop is not the bytes in flight)
  constexpr int DEPTH = 1;
  const int64_t grid = gridDim.x;
  const int64_t pEnd = r.phase == 1 ? 0 : r.numParts;
  // the workgroup's i-th partition: partition (i mod G) of its (i / G)-th group of partitions
  const int64_t numGroups = (pEnd + G - 1) >> gs;
  const int64_t myGroups = blockIdx.x < numGroups ? (numGroups - blockIdx.x + grid - 1) / grid : 0;
  const int64_t mine = myGroups << gs;
  auto partOf = [&](int64_t i) -> int64_t {
    return ((static_cast<int64_t>(blockIdx.x) + (i >> gs) * grid) << gs) + (i & (G - 1));
  };
  uint64_t rangeBegin[DEPTH + 1], rangeEnd[DEPTH + 1];
  uint64_t ahead[DEPTH][kHashAhead][W];
#pragma unroll
  for (int d = 0; d <= DEPTH; ++d) {
    rangeBegin[d] = rangeEnd[d] = 0;
    if (d < mine && partOf(d) < pEnd) {
      rpPartitionRangeAhead(r, partOf(d), &rangeBegin[d], &rangeEnd[d]);
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    hashFoldLoadAhead<W, KR>(r, rangeBegin[d], rangeEnd[d], ahead[d]);
  }
  bool groupHas = false, groupSplit = false;
  for (int64_t i = 0; i < mine; ++i) {
    const int64_t p = partOf(i);
    const uint64_t begin = rangeBegin[0], end = rangeEnd[0];
    uint64_t w[kHashAhead][W];
#pragma unroll
    for (int u = 0; u < kHashAhead; ++u) {
#pragma unroll
      for (int q = 0; q < W; ++q) {
        w[u][q] = ahead[0][u][q];
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; ++d) {
          ahead[d][u][q] = ahead[d + 1][u][q];
        }
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      rangeBegin[d] = rangeBegin[d + 1];
      rangeEnd[d] = rangeEnd[d + 1];
    }
    rangeBegin[DEPTH] = rangeEnd[DEPTH] = 0;
    if (i + DEPTH + 1 < mine && partOf(i + DEPTH + 1) < pEnd) {
      rpPartitionRangeAhead(r, partOf(i + DEPTH + 1), &rangeBegin[DEPTH], &rangeEnd[DEPTH]);
    }
    if (i + DEPTH < mine) {
      hashFoldLoadAhead<W, KR>(r, rangeBegin[DEPTH - 1], rangeEnd[DEPTH - 1], ahead[DEPTH - 1]);
    }
    const bool lastOfGroup = (i & (G - 1)) == G - 1;
    if (end != begin) {  // uniform per workgroup
      const bool split = end - begin > r.sliceRecs;
      const uint64_t stop = split ? begin + r.sliceRecs : end;
      const uint64_t base = static_cast<uint64_t>(p >> gs << gs) << r.shiftB;
      if (split && threadIdx.x == 0) {
        r.splitList[16 + atomicAdd(&r.splitList[0], 1u)] = static_cast<uint32_t>(p);
      }
#pragma unroll
      for (int u = 0; u < kHashAhead; ++u) {
        if (begin + u * 512 + threadIdx.x < stop) {
          hashFoldRecord<W, DENSE, KR>(f, r, base, w[u]);
        }
      }
      if (stop - begin > kHashAhead * 512) {
        hashFoldRecords<W, DENSE, KR>(f, r, base, begin + kHashAhead * 512, stop);
      }
      groupHas = true;
      groupSplit = groupSplit || split;
    }
    if (lastOfGroup && groupHas) {
      blockSync();  // every record of the group is in the table
      if constexpr (DENSE) {
        hashFoldFlushDense(f, r, !groupSplit, &parity, &newGroups);
      } else {
        hashFoldFlush(f, r, !groupSplit);
      }
      groupHas = groupSplit = false;
    }
  }
  // Remaining slices of the split partitions (skewed keys), in a launch of its own (r.splitList is
  // complete): folded by all workgroups, flushed with atomics.
  const uint32_t numSplit = r.phase == 1 ? r.splitList[0] : 0;
  for (uint32_t q = 0; q < numSplit; ++q) {
    const int64_t p = r.splitList[16 + q];
    uint64_t begin, end;
    rpPartitionRange(r, p, &begin, &end);
    const uint64_t base = static_cast<uint64_t>(p >> gs << gs) << r.shiftB;
    const uint64_t slices = (end - begin + r.sliceRecs - 1) / r.sliceRecs;
    for (uint64_t sl = 1 + blockIdx.x; sl < slices; sl += gridDim.x) {
      const uint64_t b = begin + sl * r.sliceRecs;
      hashFoldRecords<W, DENSE, KR>(f, r, base, b, b + r.sliceRecs < end ? b + r.sliceRecs : end);
      if constexpr (DENSE) {
        hashFoldFlushDense(f, r, false, &parity, &newGroups);
      } else {
        hashFoldFlush(f, r, false);
      }
    }
  }
  if constexpr (DENSE) {
    // what is left of the workgroup's last block stays empty; its groups are counted once
    blockSync();
    denseFillHoles(r, alloc[DA_CURSOR + parity], alloc[DA_LIMIT + parity]);
    if (threadIdx.x == 0 && newGroups != 0) {
      atomicAdd(&r.counters->numNewGroups, newGroups);
    }
  }
}

// How many distinct keys does a partition of a hashed level-2 layout hold? gridDim.x partitions
// spread over the layout, one workgroup each, the keys of up to 2048 records into an LDS set:
// out[0] += distinct keys, out[1] += records looked at, out[2] = max distinct keys of one partition.
// (Partitions are ranges of the key HASH: their numbers of distinct keys are alike whatever the
// keys' frequencies are - the launch sizes the folds' LDS tables from this, see hashSlots.)
template <int W>
__global__ __launch_bounds__(256) void k_rp_distinct_sample(const uint64_t* recs, const uint64_t* partBase,
                                                            const uint32_t* partCount, int64_t numParts, uint32_t* out,
                                                            int keyWord) {
  constexpr int kSet = 4096;
  __shared__ unsigned long long set[kSet];
  __shared__ uint32_t found;
  for (int i = threadIdx.x; i < kSet; i += blockDim.x) {
    set[i] = kEmpty;
  }
  if (threadIdx.x == 0) {
    found = 0;
  }
  blockSync();
  const int64_t p = static_cast<int64_t>(blockIdx.x) * (numParts / gridDim.x);
  const uint64_t begin = partBase[p];
  const uint32_t n = partCount[p] < 2048u ? partCount[p] : 2048u;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const unsigned long long key = recs[(begin + i) * W + keyWord];
    uint32_t pos = static_cast<uint32_t>(twangMix64(key) >> 20) & (kSet - 1);
    for (;;) {
      const unsigned long long k = set[pos];
      if (k == key) {
        break;
      }
      if (k == kEmpty) {
        const unsigned long long old = atomicCAS(&set[pos], kEmpty, key);
        if (old == kEmpty) {
          atomicAdd(&found, 1u);
          break;
        }
        if (old == key) {
          break;
        }
      }
      pos = (pos + 1) & (kSet - 1);
    }
  }
  blockSync();
  if (threadIdx.x == 0) {
    atomicAdd(&out[0], found);
    atomicAdd(&out[1], n);
    atomicMax(&out[2], found);
  }
}

// The same question BEFORE level 2 (round 5): how many distinct keys would a partition hold? Level 1 has put the
// records of 2^shift2 consecutive partitions into one bucket; kBucketSamples buckets are scanned completely
// (kBucketSlices workgroups each) and the records of ONE of their partitions - (home slot >> shiftB) & (2^shift2 - 1)
// == 0 - are counted exactly: their keys go into a set per bucket in HBM (~1000 inserts per bucket). A sample of the
// input rows could not tell ten rows per key from one at 10^8 keys; all records of a slot range can. The launch
// then moves bits from shift2 to shiftB: fewer, larger partitions whose keys still fit the folds' LDS table, i.e.
// fewer bins at level 2 and longer runs per bin and sub-tile (config 4 with sparse keys: 1024 -> 256 bins).
constexpr int kBucketSamples = 32;
constexpr int kBucketSlices = 16;
constexpr int kBucketSet = 4096;
template <int W, bool KR>
__global__ __launch_bounds__(1024) void k_rp_bucket_sample(const uint64_t* recs, const uint64_t* binFirst,
                                                            const uint32_t* binCursor, int32_t numBins, int32_t shiftB,
                                                            int32_t shift2, uint64_t slotMask, uint64_t keyMask,
                                                            unsigned long long* sets, uint32_t* perBucket) {
  const int sb = blockIdx.x / kBucketSlices, slice = blockIdx.x % kBucketSlices;
  const int bucket = static_cast<int>((static_cast<int64_t>(sb) * numBins) / kBucketSamples);
  const uint64_t begin = binFirst[bucket];
  const uint32_t count = binCursor[bucket];
  const uint32_t per = (count + kBucketSlices - 1) / kBucketSlices;
  const uint32_t lo = slice * per;
  const uint32_t hi = lo + per < count ? lo + per : count;
  unsigned long long* set = sets + static_cast<size_t>(sb) * kBucketSet;
  const uint64_t binMask = (1ULL << shift2) - 1;
  uint32_t mineRecords = 0, mineFound = 0;
  constexpr int kAhead = 8;   // loads in flight per lane (a scan of 16 - 24 MB per bucket: latency, not bytes)
  for (uint32_t at = lo; at < hi; at += kAhead * blockDim.x) {
    uint64_t w0s[kAhead];
    unsigned long long keys[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const uint32_t i = at + u * blockDim.x + threadIdx.x;
      const uint64_t rec = begin + (i < hi ? i : hi - 1);   // clamped, unconditional
      w0s[u] = recs[rec * W];
      keys[u] = KR ? w0s[u] : recs[rec * W + (W - 1)];
    }
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const uint32_t i = at + u * blockDim.x + threadIdx.x;
      const unsigned long long key = keys[u];
      const uint64_t part = KR ? (twangMix64(key) & slotMask) : (w0s[u] & keyMask);
      if (i >= hi || ((part >> shiftB) & binMask) != 0) {
        continue;
      }
      ++mineRecords;
      uint32_t pos = static_cast<uint32_t>(twangMix64(key) >> 20) & (kBucketSet - 1);
      for (int probes = 0; probes < kBucketSet; ++probes) {
        const unsigned long long old = atomicCAS(&set[pos], static_cast<unsigned long long>(kEmpty), key);
        if (old == kEmpty) {
          ++mineFound;
          break;
        }
        if (old == key) {
          break;
        }
        pos = (pos + 1) & (kBucketSet - 1);
      }
    }
  }
  if (mineRecords != 0) {
    atomicAdd(&perBucket[2 * sb], mineFound);
    atomicAdd(&perBucket[2 * sb + 1], mineRecords);
  }
}

// Rows appended by dense folds where one key may own several rows (skewed keys: a partition
// folded in slices): every row goes to its group row in an initialised open-addressing table
// with atomics, like the flush of a slice.
struct DenseMergeArgs {
  const uint64_t* rows;
  uint64_t numRows;
  uint64_t* table;
  uint64_t capacity;
  int32_t stride;
  int32_t numWords;
  int32_t wordKind[2 * kRadixMaxAccs];
  int32_t wordOff[2 * kRadixMaxAccs];
  Counters* counters;
};

__global__ __launch_bounds__(256) void k_dense_merge(DenseMergeArgs a) {
  for (uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; i < a.numRows;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const uint64_t* row = a.rows + i * a.stride;
    if (row[1] == kNoRow) {
      continue;  // a row no fold used (see denseFlags)
    }
    uint64_t* g = findOrInsert(a.table, a.stride, a.capacity, row[0], a.counters);
    if (g == nullptr) {
      continue;
    }
    const unsigned long long old = atomicMin(reinterpret_cast<unsigned long long*>(g + 1), row[1]);
    if (old == kNoRow) {
      atomicAdd(&a.counters->numNewGroups, 1u);
    }
    for (int j = 0; j < a.numWords; ++j) {
      const uint64_t v = row[a.wordOff[j]];
      const int32_t kind = a.wordKind[j];
      if (v != accIdentity(kind)) {
        if (kind == ACC_SUM_I64) {
          addPartial128Global(g + a.wordOff[j], v, 0);  // the row's own high word follows as word j + 1
        } else {
          applyGlobal(g + a.wordOff[j], kind == ACC_COUNT ? ACC_SUM_I64_WRAP : kind, v, a.counters);
        }
      }
    }
  }
}

// ---- generic hash mode (the reference's kHash, HashTable.cpp:470-520) ---------------
// Keys that have no 64-bit normalized form (REAL / DOUBLE / TIMESTAMP, strings
// of 8..12 bytes, key sets wider than 64 bits) are grouped through an
// open-addressing table of 8-byte slots {hash tag : 32 | group id + 1 : 32}
// probed from the VectorHasher hash (hashOne + hashMix, VectorHasher.cpp:61-126).
// A slot is claimed with one CAS (tag | PENDING); the winner takes the next dense
// group id, stores the key images (8-byte agent-scope atomic stores: readers on
// other XCDs use agent-scope atomic loads, the one form that is coherent across
// the per-XCD L2s without fences), then publishes the id. A tag match is
// confirmed by comparing the stored key images. Group rows (first input row +
// accumulators) are indexed by the dense id, so everything downstream of
// "which group row" is shared with array mode.
constexpr uint32_t kPendingGid = 0xffffffffu;
constexpr uint32_t kDeadGid = 0xfffffffeu;  // a claim that could not get a group id (table full)

struct GenericPart {
  uint64_t* slots;
  uint64_t slotMask;
  uint64_t* keyStore[kMaxKeys];   // 1 or 2 words per group
  int32_t keyWords[kMaxKeys];
  uint64_t* nullStore;            // bit k = key k is null
  uint64_t* hashStore;
  uint32_t* gidCounter;
  uint32_t maxGroups;
  uint32_t pad;
  // Arena for grouping strings longer than 12 bytes (StringView's non-inline form,
  // type/StringView.h:76-77): the group's key image keeps {size | prefix, arena pointer}.
  char* arenaBase;
  unsigned long long* arenaCursor;  // bytes handed out of the current block
  uint64_t arenaCap;
};

// Up to 8 bytes at p (n >= 1), zero padded: strings end anywhere.
__device__ inline uint64_t loadBytes8(const uint8_t* p, uint32_t n) {
  uint64_t v = 0;
  const uint32_t m = n < 8 ? n : 8;
  for (uint32_t i = 0; i < m; ++i) {
    v |= static_cast<uint64_t>(p[i]) << (8 * i);
  }
  return v;
}

// Total arena bytes the long strings of the key columns of a chunk can ask for.
struct LongBytesArgs {
  ColView keys[kMaxKeys];
  int32_t numKeys;
  int64_t numRows;
  const int32_t* rowList;
  unsigned long long* total;
};

__global__ __launch_bounds__(256) void k_long_key_bytes(LongBytesArgs a) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  unsigned long long mine = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.numRows; i += stride) {
    const int64_t row = a.rowList ? a.rowList[i] : i;
    for (int k = 0; k < a.numKeys; ++k) {
      const ColView& c = a.keys[k];
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
    atomicAdd(a.total, mine);
  }
}

struct GenericArgs {
  AggArgs a;
  GenericPart g;
  int32_t skipInside;  // rescan after a mode switch: rows inside a.keys[].range were aggregated already
  int32_t pad;
};

// True when every key of the row maps into the ranges in a.keys[].range, i.e.
// the normalized-key launch that ran before the switch consumed the row.
__device__ inline bool insideRanges(const AggArgs& a, int64_t row) {
  for (int k = 0; k < a.numKeys; ++k) {
    const KeyArg& ka = a.keys[k];
    if (colIsNull(ka.col, row)) {
      continue;
    }
    int64_t value;
    bool mappable;
    if (valueIdAt(ka.col, colIndex(ka.col, row), ka.range, &value, &mappable) == 0) {
      return false;
    }
  }
  return true;
}
static_assert(sizeof(GenericArgs) <= 4096, "kernel arguments are limited to 4 KB");

// Dense group id of one input row in generic mode (claims a slot for a new key);
// false: the row does not take part (filtered, null key dropped, unsupported key).
__device__ inline bool genericGroupId(const GenericArgs& args, int64_t row, uint32_t* gidOut) {
  const AggArgs& a = args.a;
  const GenericPart& g = args.g;
  if (a.numTerms && !evalFilter(a.terms, a.numTerms, row)) {
    return false;
  }
  if (args.skipInside && insideRanges(a, row)) {
    return false;
  }
  // Key images, null mask and the VectorHasher hash of the row.
  uint64_t w0[kMaxKeys], w1[kMaxKeys];
  uint64_t nullMask = 0;
  uint64_t hash = 0;
  bool supported = true;
#pragma unroll
  for (int k = 0; k < kMaxKeys; ++k) {
    w0[k] = 0;
    w1[k] = 0;
    if (k < a.numKeys) {
      const ColView& c = a.keys[k].col;
      uint64_t hv = kNullHash;
      if (colIsNull(c, row)) {
        nullMask |= 1ULL << k;
      } else {
        const int64_t i = colIndex(c, row);
        keyImage(c, i, &w0[k], &w1[k], &supported);  // long strings: {size | prefix, pointer}
        hv = hashValueAt(c, i);
      }
      hash = k == 0 ? hv : hashMix(hash, hv);
    }
  }
  if (nullMask && a.ignoreNullKeys) {
    return false;
  }
  // which keys of this row are non-inline strings (compared by content, stored in the arena)
  uint32_t longMask = 0;
#pragma unroll
  for (int k = 0; k < kMaxKeys; ++k) {
    if (k < a.numKeys && !((nullMask >> k) & 1) &&
        (a.keys[k].col.kind == VX355_VARCHAR || a.keys[k].col.kind == VX355_VARBINARY) &&
        static_cast<uint32_t>(w0[k]) > 12) {
      longMask |= 1u << k;
    }
  }
  const uint64_t tag = hash >> 32;
  uint64_t pos = slotOfHash(hash, g.slotMask);
  uint32_t gid = kPendingGid;
  uint64_t probes = 0;
  uint32_t spins = 0;  // every wait on another lane's publish is bounded
  while (gid == kPendingGid && probes <= g.slotMask && spins < (1u << 22)) {
    uint64_t w = loadAgent(g.slots + pos);
    bool advance = false;
    if (w == 0) {
      const unsigned long long claim = (tag << 32) | kPendingGid;
      const unsigned long long old =
          atomicCAS(reinterpret_cast<unsigned long long*>(g.slots + pos), 0ULL, claim);
      if (old == 0) {
        // Dense group ids: one atomic per wave for all lanes that claimed a slot in
        // this iteration (a single address takes < 100 M atomics/s).
        const uint64_t winners = ballot(true);
        const int leader = __ffsll(static_cast<long long>(winners)) - 1;
        uint32_t idBase = 0;
        if (lane() == leader) {
          idBase = atomicAdd(g.gidCounter, static_cast<uint32_t>(popc64(winners)));
        }
        const uint32_t id = __shfl(idBase, leader, kWave) + lanePrefix(winners);
        bool arenaOk = true;
        if (id < g.maxGroups && longMask) {
          // copy the long strings into the arena (write-through stores, like the key images)
#pragma unroll
          for (int k = 0; k < kMaxKeys; ++k) {
            if ((longMask >> k) & 1) {
              const uint32_t size = static_cast<uint32_t>(w0[k]);
              const uint32_t padded = (size + 7) & ~7u;
              const unsigned long long at = atomicAdd(g.arenaCursor, static_cast<unsigned long long>(padded));
              if (at + padded > g.arenaCap) {
                arenaOk = false;
              } else {
                const uint8_t* src = reinterpret_cast<const uint8_t*>(w1[k]);
                uint64_t* dst = reinterpret_cast<uint64_t*>(g.arenaBase + at);
                for (uint32_t off = 0; off < size; off += 8) {
                  storeAgent(dst + (off >> 3), loadBytes8(src + off, size - off));
                }
                w1[k] = reinterpret_cast<uint64_t>(dst);
              }
            }
          }
        }
        if (id < g.maxGroups && arenaOk) {
#pragma unroll
          for (int k = 0; k < kMaxKeys; ++k) {
            if (k < a.numKeys) {
              storeAgent(g.keyStore[k] + static_cast<uint64_t>(id) * g.keyWords[k], w0[k]);
              if (g.keyWords[k] == 2) {
                storeAgent(g.keyStore[k] + static_cast<uint64_t>(id) * 2 + 1, w1[k]);
              }
            }
          }
          storeAgent(g.nullStore + id, nullMask);
          storeAgent(g.hashStore + id, hash);
          // Publish after the key images: those were agent-scope (write-through)
          // atomic stores, so waiting for their acknowledgement orders them before
          // the slot store for every reader that uses agent-scope loads. A full
          // agent-scope release fence (__threadfence) also writes back the XCD's L2
          // on gfx950 and made this kernel 3x slower.
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_s_waitcnt(0);
          storeAgent(g.slots + pos, (tag << 32) | (static_cast<uint64_t>(id) + 1));
          gid = id;
        } else {
          // Never leave a PENDING slot behind: waiters would spin on it.
          storeAgent(g.slots + pos, (tag << 32) | kDeadGid);
          a.counters->tableFull = 1;
          gid = kDeadGid;  // leave the loop; the host raises the error
        }
        w = 0;
      } else {
        w = old;
      }
    }
    if (gid == kPendingGid && w != 0) {
      if ((w >> 32) == tag) {
        const uint32_t lo = static_cast<uint32_t>(w);
        if (lo == kDeadGid) {
          advance = true;
        } else if (lo == kPendingGid) {
          ++spins;  // the claimer has not published yet; look at this slot again
        } else {
          const uint32_t cand = lo - 1;
          bool equal = loadAgent(g.nullStore + cand) == nullMask;
#pragma unroll
          for (int k = 0; k < kMaxKeys; ++k) {
            if (equal && k < a.numKeys && !((nullMask >> k) & 1)) {
              equal = loadAgent(g.keyStore[k] + static_cast<uint64_t>(cand) * g.keyWords[k]) == w0[k];
              if (equal && g.keyWords[k] == 2) {
                const uint64_t stored = loadAgent(g.keyStore[k] + static_cast<uint64_t>(cand) * 2 + 1);
                if ((longMask >> k) & 1) {
                  // same size and prefix: compare the bytes (the group's copy is zero padded to 8)
                  const uint32_t size = static_cast<uint32_t>(w0[k]);
                  const uint64_t* theirs = reinterpret_cast<const uint64_t*>(stored);
                  const uint8_t* mine = reinterpret_cast<const uint8_t*>(w1[k]);
                  for (uint32_t off = 0; off < size && equal; off += 8) {
                    equal = loadAgent(theirs + (off >> 3)) == loadBytes8(mine + off, size - off);
                  }
                } else {
                  equal = stored == w1[k];
                }
              }
            }
          }
          if (equal) {
            gid = cand;
          } else {
            advance = true;
          }
        }
      } else {
        advance = true;
      }
    }
    if (advance) {
      pos = (pos + 1) & g.slotMask;
      ++probes;
    }
  }
  if (gid == kPendingGid || gid == kDeadGid) {
    a.counters->tableFull = 1;
    return false;
  }
  *gidOut = gid;
  return true;
}

__global__ __launch_bounds__(256) void k_agg_generic(GenericArgs args) {
  const AggArgs& a = args.a;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t rounds = (a.numRows + stride - 1) / stride;
  uint32_t newGroups = 0;
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (int64_t r = 0; r < rounds; ++r, i += stride) {
    bool active = false;
    int64_t row = 0;
    uint32_t gid = 0;
    if (i < a.numRows) {
      row = a.rowList ? a.rowList[i] : i;
      active = genericGroupId(args, row, &gid);
    }
    // the group row is table + gid * stride (mode = array): hot groups are combined per wave
    updateGlobalWave(a, row, gid, active, &newGroups);
  }
  addNewGroups(a.counters, newGroups);
}

// Re-inserts every group id into a larger slot array (HashTable::rehash).
struct ToGenericArgs {
  const uint64_t* oldTable;
  uint64_t oldRows;
  int32_t oldMode;
  int32_t stride;
  int32_t numKeys;
  int32_t pad;
  KeyRange range[kMaxKeys];
  int32_t kind[kMaxKeys];
  uint64_t* newTable;
  GenericPart g;
};

// Mode switch normalized key / array -> generic (the reference re-decides the
// hash mode and rehashes when a VectorHasher can no longer produce value ids,
// HashTable.cpp:1751-1839): every live group gets a dense id, its keys are
// decoded from the normalized key into the stored images + VectorHasher hash the
// generic kernel compares against, and its row words move to newTable[id].
__global__ __launch_bounds__(256) void k_to_generic(ToGenericArgs a) {
  const uint64_t step = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  const uint64_t rounds = (a.oldRows + step - 1) / step;
  uint64_t r = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (uint64_t it = 0; it < rounds; ++it, r += step) {
    const uint64_t* src = a.oldTable + r * a.stride;
    const bool live = r < a.oldRows && src[1] != kNoRow;
    const uint64_t m = ballot(live);
    if (m == 0) {
      continue;
    }
    const int leader = __ffsll(static_cast<long long>(m)) - 1;
    uint32_t base = 0;
    if (lane() == leader) {
      base = atomicAdd(a.g.gidCounter, static_cast<uint32_t>(popc64(m)));
    }
    base = __shfl(base, leader, kWave);
    if (!live) {
      continue;
    }
    const uint32_t id = base + lanePrefix(m);
    const uint64_t key = a.oldMode == MODE_ARRAY ? r : src[0];
    uint64_t nullMask = 0;
    uint64_t hash = 0;
    for (int k = 0; k < a.numKeys; ++k) {
      const uint64_t vid = (key / a.range[k].multiplier) % a.range[k].rangeSize;
      uint64_t w0 = 0, w1 = 0, hv = kNullHash;
      if (vid == 0) {
        nullMask |= 1ULL << k;
      } else if (a.kind[k] == VX355_BOOLEAN) {
        w0 = vid == 2 ? 1 : 0;
        hv = vid == 2 ? ~0ULL : 0ULL;
      } else {
        const int64_t v = static_cast<int64_t>(vid - 1 + static_cast<uint64_t>(a.range[k].min));
        if (a.kind[k] == VX355_VARCHAR || a.kind[k] == VX355_VARBINARY) {
          // Inverse of stringAsNumber: the marker bit sits right above the bytes;
          // 0 is the empty string.
          uint32_t size = 0;
          uint64_t bytes = 0;
          if (v != 0) {
            const int top = 63 - __clzll(static_cast<long long>(v));
            size = static_cast<uint32_t>(top >> 3);
            bytes = static_cast<uint64_t>(v) - (1ULL << top);
          }
          w0 = static_cast<uint64_t>(size) | ((bytes & 0xffffffffULL) << 32);
          w1 = bytes >> 32;
          uint8_t buf[8];
#pragma unroll
          for (int b = 0; b < 8; ++b) {
            buf[b] = static_cast<uint8_t>(bytes >> (8 * b));
          }
          hv = hashBytes(1, buf, static_cast<int32_t>(size));
        } else if (a.kind[k] == VX355_BIGINT) {
          w0 = static_cast<uint64_t>(v);
          hv = twangMix64(static_cast<uint64_t>(v));
        } else {
          w0 = static_cast<uint64_t>(v);
          hv = jenkinsRevMix32(static_cast<uint32_t>(static_cast<int32_t>(v)));
        }
      }
      a.g.keyStore[k][static_cast<uint64_t>(id) * a.g.keyWords[k]] = w0;
      if (a.g.keyWords[k] == 2) {
        a.g.keyStore[k][static_cast<uint64_t>(id) * 2 + 1] = w1;
      }
      hash = k == 0 ? hv : hashMix(hash, hv);
    }
    a.g.nullStore[id] = nullMask;
    a.g.hashStore[id] = hash;
    uint64_t* dst = a.newTable + static_cast<uint64_t>(id) * a.stride;
    for (int w = 1; w < a.stride; ++w) {
      dst[w] = src[w];
    }
  }
}

__global__ __launch_bounds__(256) void k_generic_rehash(uint64_t* slots, uint64_t slotMask,
                                                         const uint64_t* hashStore, uint32_t numGroups) {
  const uint32_t step = gridDim.x * blockDim.x;
  for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < numGroups; id += step) {
    const uint64_t hash = hashStore[id];
    const unsigned long long word = ((hash >> 32) << 32) | (static_cast<uint64_t>(id) + 1);
    uint64_t pos = slotOfHash(hash, slotMask);
    while (atomicCAS(reinterpret_cast<unsigned long long*>(slots + pos), 0ULL, word) != 0) {
      pos = (pos + 1) & slotMask;
    }
  }
}

// ---- key statistics of the first rows (VectorHasher::analyze) ---------------
struct StatsArgs {
  KeyArg keys[kMaxKeys];
  int32_t numKeys;
  int64_t numRows;
  Counters* counters;
};

// Distinct normalized keys among 'a.numRows' rows taken every 'step' rows of the batch (one
// workgroup, an LDS hash set of 8192 entries): out[0] = the count, saturating near 4096.
constexpr int kCardSetSize = 8192;
__global__ __launch_bounds__(1024) void k_card_sample(AggArgs a, int64_t step, uint32_t* set, uint32_t* out) {
  // The sampled rows lie ~n / 16384 rows apart: every one of them is a cold line AND a cold page in
  // each key column, so the pass is bound by translation misses per lane. 16 workgroups share the
  // work (one or two rows per lane) and the set - 8192 words in HBM, filled with 0xff by the host;
  // out[0] counts the insertions (one workgroup with an LDS set took 0.21 ms on TPC-H Q1).
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.numRows; i += stride) {
    uint64_t key;
    if (normalizedKey(a, i * step, &key) != 0) {
      continue;  // (dropped row, or a key outside the ranges the first statistics pass found)
    }
    if (__hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= static_cast<uint32_t>(kCardSetSize / 2)) {
      continue;  // saturated
    }
    // (open-addressing tables: 64-bit keys; 32 mixed bits tell them apart well enough for an estimate)
    const uint32_t k32 = static_cast<uint32_t>(twangMix64(key) >> 7) & 0x7fffffffu;
    uint32_t pos = static_cast<uint32_t>((key * 0x9E3779B97F4A7C15ULL) >> 40) & (kCardSetSize - 1);
    for (int probes = 0; probes < 64; ++probes) {
      const uint32_t seen = __hip_atomic_load(&set[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (seen == k32) {
        break;
      }
      if (seen == 0xffffffffu) {
        const uint32_t old = atomicCAS(&set[pos], 0xffffffffu, k32);
        if (old == 0xffffffffu) {
          atomicAdd(out, 1u);
          break;
        }
        if (old == k32) {
          break;
        }
      }
      pos = (pos + 1) & (kCardSetSize - 1);
    }
  }
}

// Distinct keys among the first kFirstRowsProbe rows (block 0 only; an LDS hash set over a mix of the
// keys' int64 images): with a few hundred of them among so few rows, lane-shared LDS accumulators
// will not pile up on a handful of addresses, and a direct LDS layout can be chosen for the whole
// first batch without the small probing chunk that otherwise counts the live groups first.
constexpr int kFirstRowsProbe = 2048;
template <typename Args>
__device__ inline void firstRowsDistinct(const Args& keys, int32_t numKeys, int64_t numRows, Counters* counters) {
  __shared__ uint64_t set[2 * kFirstRowsProbe];
  __shared__ uint32_t distinct;
  for (int i = threadIdx.x; i < 2 * kFirstRowsProbe; i += blockDim.x) {
    set[i] = ~0ULL;
  }
  if (threadIdx.x == 0) {
    distinct = 0;
  }
  blockSync();
  const int64_t rows = numRows < kFirstRowsProbe ? numRows : kFirstRowsProbe;
  // every thread first fetches ALL its rows' keys (independent loads: one memory latency, not
  // eight in a row behind the set's atomics), then inserts their hashes
  constexpr int kPerThread = kFirstRowsProbe / 256;
  uint64_t hs[kPerThread];
#pragma unroll
  for (int u = 0; u < kPerThread; ++u) {
    const int64_t row = static_cast<int64_t>(threadIdx.x) + static_cast<int64_t>(u) * blockDim.x;
    uint64_t h = 0x51ed270b27b4f3cfULL;
    if (row < rows) {
      for (int k = 0; k < numKeys; ++k) {
        const ColView& c = keys.keys[k].col;
        uint64_t v = 0x7ff8dead00000000ULL;  // null
        if (!colIsNull(c, row)) {
          KeyRange all;
          all.min = INT64_MIN;
          all.max = INT64_MAX;
          int64_t id;
          bool mappable;
          valueIdAt(c, colIndex(c, row), all, &id, &mappable);
          v = static_cast<uint64_t>(id);
        }
        h = hashMix(h, v);
      }
    }
    hs[u] = row < rows ? (h & ~(1ULL << 63)) : ~0ULL;  // (never the empty marker)
  }
#pragma unroll
  for (int u = 0; u < kPerThread; ++u) {
    const uint64_t h = hs[u];
    if (h == ~0ULL) {
      continue;
    }
    uint32_t pos = static_cast<uint32_t>(h >> 20) & (2 * kFirstRowsProbe - 1);
    for (int probes = 0; probes < 2 * kFirstRowsProbe; ++probes) {
      const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&set[pos]), ~0ULL, h);
      if (old == ~0ULL) {
        atomicAdd(&distinct, 1u);
        break;
      }
      if (old == h) {
        break;
      }
      pos = (pos + 1) & (2 * kFirstRowsProbe - 1);
    }
  }
  blockSync();
  if (threadIdx.x == 0) {
    counters->firstRowsDistinct = distinct;
  }
}

// (Args = the kernel's own argument struct, StatsArgs or AggArgs: the key views are read straight
// from the kernel arguments - a pointer into them would turn every access into a flat load.)
template <typename Args>
__device__ inline void keyStatsBody(const Args& args, int64_t numRows, int block, int numBlocks) {
  struct {
    const Args& k;
    int32_t numKeys;
    int64_t numRows;
    Counters* counters;
  } a{args, args.numKeys, numRows, args.counters};
  // (the LAST block of the launch only probes the first rows: it runs next to the others instead of
  // in front of one of them)
  if (block == numBlocks) {
    firstRowsDistinct(args, args.numKeys, numRows, args.counters);
    return;
  }
  const int64_t stride = static_cast<int64_t>(numBlocks) * blockDim.x;
  int64_t mn[kMaxKeys], mx[kMaxKeys];
  for (int k = 0; k < kMaxKeys; ++k) {
    mn[k] = INT64_MAX;
    mx[k] = INT64_MIN;
  }
  bool unmappable = false;
  for (int64_t row = static_cast<int64_t>(block) * blockDim.x + threadIdx.x; row < a.numRows;
       row += stride) {
    for (int k = 0; k < a.numKeys; ++k) {
      const ColView& c = a.k.keys[k].col;
      if (colIsNull(c, row)) {
        continue;
      }
      KeyRange all;
      all.min = INT64_MIN;
      all.max = INT64_MAX;
      int64_t v;
      bool mappable;
      valueIdAt(c, colIndex(c, row), all, &v, &mappable);
      if (!mappable) {
        unmappable = true;
        continue;
      }
      mn[k] = v < mn[k] ? v : mn[k];
      mx[k] = v > mx[k] ? v : mx[k];
    }
  }
  // One pair of atomics per WORKGROUP (a single address takes < 100 M atomics per second: with one
  // pair per wave a whole 8 M-row batch spent 0.3 ms here): waves reduce with shuffles, then LDS.
  __shared__ int64_t waveLo[4][kMaxKeys];
  __shared__ int64_t waveHi[4][kMaxKeys];
  for (int k = 0; k < a.numKeys; ++k) {
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
  if (threadIdx.x < static_cast<unsigned>(a.numKeys)) {
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
  if (unmappable) {
    a.counters->unmappable = 1;
  }
}

__global__ __launch_bounds__(256) void k_key_stats(StatsArgs a) {
  keyStatsBody(a, a.numRows, blockIdx.x, static_cast<int>(gridDim.x) - 1);
}

// Largest |input| of every DOUBLE sum over the analysed prefix: fixes the grid
// of the hi/lo split (AccArg::splitM).
__device__ inline void sumStatsBody(const AggArgs& a, int64_t numRows, int block, int numBlocks) {
  // One accumulator after the other (a few rows per thread: re-reading them per DOUBLE sum is
  // nothing) - sixteen unrolled copies of accInput needed 332 registers and 3.5 KB of scratch per
  // lane, which every block of k_first_stats paid for, the key-statistics blocks included.
  const int64_t stride = static_cast<int64_t>(numBlocks) * blockDim.x;
  for (int j = 0; j < a.numAccs; ++j) {
    if (a.accs[j].kind != ACC_SUM_F64) {
      continue;
    }
    uint64_t m = 0;
    for (int64_t row = static_cast<int64_t>(block) * blockDim.x + threadIdx.x; row < numRows; row += stride) {
      if (a.numTerms && !evalFilter(a.terms, a.numTerms, row)) {
        continue;
      }
      uint64_t v;
      if (accInput(a, a.accs[j], row, &v)) {
        v &= 0x7fffffffffffffffULL;  // |v| as a bit pattern orders like the magnitude
        m = v > m ? v : m;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint64_t o = shfl64(m, lane() ^ off);
      m = o > m ? o : m;
    }
    if (lane() == 0 && m != 0) {
      atomicMax(reinterpret_cast<unsigned long long*>(&a.counters->sumMax[j]), m);
    }
  }
}

__global__ __launch_bounds__(256) void k_sum_stats(AggArgs a) {
  sumStatsBody(a, a.numRows, blockIdx.x, gridDim.x);
}

// Both statistics passes of an operator's first batch in ONE launch (two tiny latency-bound kernels
// in a row cost more in launch gaps than in work): blocks [0, keyBlocks) analyse the keys of the
// first keyRows rows, the others the DOUBLE sums' inputs of the first a.numRows rows.
__global__ __launch_bounds__(256) void k_first_stats(AggArgs a, int64_t keyRows, int32_t keyBlocks, CounterMail mail) {
  if (static_cast<int32_t>(blockIdx.x) <= keyBlocks) {
    keyStatsBody(a, keyRows, blockIdx.x, keyBlocks);  // block keyBlocks: the distinct probe
  } else {
    sumStatsBody(a, a.numRows, static_cast<int>(blockIdx.x) - keyBlocks - 1,
                 static_cast<int>(gridDim.x) - keyBlocks - 1);
  }
  blockSync();
  publishCountersFromLastBlock(mail, gridDim.x);
}

// ---- table maintenance ---------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_table(uint64_t* table, uint64_t rows, int32_t stride,
                                                     const uint64_t* pattern) {
  const uint64_t total = rows * stride;
  const uint64_t step = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  // (write-only: 4.3 GB in 2.2 ms = 1.9 TB/s whether the pattern word is found with a modulo per
  // store or incrementally - the chip's fill rate, not the index arithmetic, is the limit)
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += step) {
    table[i] = pattern[i % stride];
  }
}

struct RekeyArgs {
  const uint64_t* oldTable;
  uint64_t oldRows;
  int32_t oldMode;
  uint64_t* newTable;
  uint64_t newCapacity;
  int32_t newMode;
  int32_t stride;
  int32_t numKeys;
  KeyRange oldRange[kMaxKeys];
  KeyRange newRange[kMaxKeys];
  Counters* counters;
};

// Moves every live group to its place under the new key ranges / capacity
// (the device analogue of HashTable::rehash, HashTable.cpp:1541-1596).
__global__ __launch_bounds__(256) void k_rekey(RekeyArgs a) {
  const uint64_t step = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t r = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; r < a.oldRows;
       r += step) {
    const uint64_t* src = a.oldTable + r * a.stride;
    if (src[1] == kNoRow) {
      continue;
    }
    const uint64_t oldKey = a.oldMode == MODE_ARRAY ? r : src[0];
    uint64_t newKey = 0;
    for (int k = 0; k < a.numKeys; ++k) {
      uint64_t id = (oldKey / a.oldRange[k].multiplier) % a.oldRange[k].rangeSize;
      if (id != 0) {
        // value = id - 1 + oldMin ; new id = value - newMin + 1 (wraps safely)
        uint64_t nid = id + static_cast<uint64_t>(a.oldRange[k].min) -
            static_cast<uint64_t>(a.newRange[k].min);
        newKey += a.newRange[k].multiplier * nid;
      }
    }
    uint64_t* dst;
    if (a.newMode == MODE_ARRAY) {
      dst = a.newTable + newKey * a.stride;
    } else {
      dst = findOrInsert(a.newTable, a.stride, a.newCapacity, newKey, a.counters);
      if (!dst) {
        continue;
      }
    }
    for (int w = 1; w < a.stride; ++w) {
      dst[w] = src[w];
    }
  }
}

// Group rows of 'oldStride' words -> rows of 'newStride' (> oldStride) words; new words = pattern.
__global__ __launch_bounds__(256) void k_restride(const uint64_t* src, uint64_t* dst, uint64_t rows, int32_t oldStride,
                                                   int32_t newStride, const uint64_t* pattern) {
  const uint64_t total = rows * newStride;
  const uint64_t step = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += step) {
    const uint64_t r = i / newStride;
    const int32_t w = static_cast<int32_t>(i % newStride);
    dst[i] = w < oldStride ? src[r * oldStride + w] : pattern[w];
  }
}

// srcOff == 1 (the first-row word): dst = group exists ? 1 : 0 — used when a
// count that was only ever needed as a "seen" flag has to become a real word.
__global__ __launch_bounds__(256) void k_copy_acc(uint64_t* table, uint64_t rows, int32_t stride,
                                                   int32_t srcOff, int32_t dstOff) {
  const uint64_t step = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t r = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; r < rows;
       r += step) {
    const uint64_t v = table[r * stride + srcOff];
    table[r * stride + dstOff] = srcOff == 1 ? (v != kNoRow ? 1 : 0) : v;
  }
}

// Live groups -> (first row, group row index) pairs, unordered.
__global__ __launch_bounds__(256) void k_collect(const uint64_t* table, uint64_t rows, int32_t stride,
                                                  uint64_t* firstOut, uint32_t* indexOut,
                                                  uint32_t* cursor) {
  // One wave owns 32 x 64 consecutive group rows and claims its output range
  // with ONE atomic (a single HBM address takes ~88 M atomics/s, so one atomic
  // per 64 rows would dominate at 10^8 groups).
  constexpr int kGroups = 32;
  const uint64_t waveRows = 64ULL * kGroups;
  const uint64_t numWaveTiles = (rows + waveRows - 1) / waveRows;
  const uint64_t waveStride = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
  for (uint64_t t = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; t < numWaveTiles;
       t += waveStride) {
    uint64_t first[kGroups];
    uint32_t total = 0;
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      const uint64_t r = t * waveRows + static_cast<uint64_t>(g) * 64 + lane();
      first[g] = r < rows ? table[r * stride + 1] : kNoRow;
    }
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      total += popc64(ballot(first[g] != kNoRow));
    }
    if (total == 0) {
      continue;
    }
    uint32_t base = 0;
    if (lane() == 0) {
      base = atomicAdd(cursor, total);
    }
    base = __shfl(base, 0, kWave);
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      const bool liveRow = first[g] != kNoRow;
      const uint64_t m = ballot(liveRow);
      if (liveRow) {
        const uint32_t p = base + lanePrefix(m);
        firstOut[p] = first[g];
        indexOut[p] = static_cast<uint32_t>(t * waveRows + static_cast<uint64_t>(g) * 64 + lane());
      }
      base += popc64(m);
    }
  }
}

// ---- first-seen order of many groups (config 4: 10^8 of them) -----------------------------------
// A generic LSD pair sort moves 12 bytes per entry four times (3.4 ms per 10^8 entries with the vendor
// library's, measured in round 4). The entries
// are special: the keys are FIRST INPUT ROWS - distinct, because a row belongs to one group, and
// below 2^bits (bits = those of the operator's input row count). So: pack an entry into one word
// {group row : 32 | first row : 32} (k_fs_pack, in place; an entry no group owns - the holes of a
// dense fold, key ~0 - gets a row number of its own behind the input, so that it sorts last and
// the keys stay distinct), two MSD levels of 1024 bins with the radix path's own exact passes
// (k_rp_tiles + k_rp_count2 + scan + k_rp_scatter2_sorted<1>: the first rows of random keys crowd
// at the front of the input, regions of an even share would overflow), and then no third scatter:
// a partition of the second level spans 2^(bits - 20) <= 4096 consecutive row numbers, so one wave
// sets a bit per entry in an LDS bitmap and an entry's rank inside the partition is the number of
// bits below its own (k_fs_rank). Measured at 10^8 entries: 3.1 ms (pack 0.4, levels 2 x 0.9, ranks
// 0.7, tiles 0.1) against 3.4 - the 8-byte scatters reach 2.3 TB/s only. (One level and a
// workgroup per bin with the bin's whole 2^20-bit bitmap in LDS: 3.3 ms for the ranks alone - their
// stores scatter over 4 MB per workgroup.)
__global__ __launch_bounds__(256) void k_fs_pack(uint64_t* keys, const uint32_t* vals, uint64_t n, uint32_t firstFree) {
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const uint64_t first = keys[i];
    // an entry no group owns: a row number of its own behind the input (its place in the list)
    const uint32_t f = first >= firstFree ? firstFree + static_cast<uint32_t>(i) : static_cast<uint32_t>(first);
    keys[i] = (static_cast<uint64_t>(vals[i]) << 32) | f;
  }
}

__device__ inline void waveSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kFsMaxLowBits = 12;   // bitmap of one partition: 4096 bits = 64 words, one per lane

__global__ __launch_bounds__(256) void k_fs_rank(const uint64_t* recs, const uint64_t* offsets, const uint32_t* partCell,
                                                 int64_t numParts, int32_t lowBits, uint32_t* order, uint32_t* error) {
  __shared__ unsigned long long bitmap[4][64];
  __shared__ uint32_t below[4][64];
  const int wave = threadIdx.x >> 6;
  const int l = lane();
  const int words = lowBits <= 6 ? 1 : 1 << (lowBits - 6);
  const uint32_t lowMask = (1u << lowBits) - 1;
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * 4 + wave; p < numParts; p += static_cast<int64_t>(gridDim.x) * 4) {
    const uint64_t begin = offsets[partCell[p]];
    const uint64_t end = offsets[partCell[p + 1]];
    if (begin == end) {
      continue;  // uniform per wave
    }
    if (l < words) {
      bitmap[wave][l] = 0;
    }
    waveSync();
    for (uint64_t i = begin + l; i < end; i += 64) {
      const uint32_t b = static_cast<uint32_t>(recs[i]) & lowMask;
      const unsigned long long bit = 1ULL << (b & 63);
      if (atomicOr(&bitmap[wave][b >> 6], bit) & bit) {
        *error = 1;  // two entries with one first row: cannot happen
      }
    }
    waveSync();
    const uint32_t mine = l < words ? static_cast<uint32_t>(popc64(bitmap[wave][l])) : 0;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(incl, off, kWave);
      if (l >= off) {
        incl += o;
      }
    }
    below[wave][l] = incl - mine;
    waveSync();
    for (uint64_t i = begin + l; i < end; i += 64) {
      const uint64_t w = recs[i];
      const uint32_t b = static_cast<uint32_t>(w) & lowMask;
      const uint64_t pos = begin + below[wave][b >> 6] +
          static_cast<uint32_t>(popc64(bitmap[wave][b >> 6] & ((1ULL << (b & 63)) - 1)));
      order[pos] = static_cast<uint32_t>(w >> 32);
    }
    waveSync();
  }
}

// Small tables (BASELINE configs 1 and 2: a few to a few thousand groups): live rows collected and
// put into first-seen order without k_collect + a count read-back + a device-wide radix sort: no
// stream synchronisation between noMoreInput and the output page. Every workgroup lists ALL live
// rows in its LDS (the table is a few KB; the list is the same in every workgroup: rows in table
// order) and ranks 64 of them by counting the entries with a smaller first-row word (they are
// distinct: a row belongs to one group): lane = one of the 64 entries x one of 16 segments of the
// list, so a thousand groups are ranked by 16 workgroups in ~60 LDS reads per lane (one workgroup
// ranking all of them alone took 20 us: 16 000 broadcast reads on one CU).
constexpr int kSmallSortMax = 4096;
__global__ __launch_bounds__(1024) void k_collect_sort_small(const uint64_t* table, uint32_t rows, int32_t stride,
                                                              uint32_t* orderOut, uint32_t capacityOut, uint32_t* found) {
  __shared__ uint64_t keys[kSmallSortMax];
  __shared__ uint32_t vals[kSmallSortMax];
  __shared__ uint32_t waveCount[16];
  __shared__ uint32_t rank[64];
  uint32_t n = 0;
  for (uint32_t base = 0; base < rows; base += blockDim.x) {
    const uint32_t r = base + threadIdx.x;
    const uint64_t first = r < rows ? table[static_cast<uint64_t>(r) * stride + 1] : kNoRow;
    const bool liveRow = first != kNoRow;
    const uint64_t m = ballot(liveRow);
    if (lane() == 0) {
      waveCount[threadIdx.x >> 6] = static_cast<uint32_t>(popc64(m));
    }
    blockSync();
    uint32_t before = n;
    for (int w = 0; w < 16; ++w) {
      const uint32_t c = waveCount[w];
      before += w < static_cast<int>(threadIdx.x >> 6) ? c : 0;
      n += c;
    }
    const uint32_t at = before + lanePrefix(m);
    if (liveRow && at < static_cast<uint32_t>(kSmallSortMax)) {
      keys[at] = first;
      vals[at] = r;
    }
    blockSync();
  }
  const uint32_t count = n;
  n = n < static_cast<uint32_t>(kSmallSortMax) ? n : static_cast<uint32_t>(kSmallSortMax);
  if (threadIdx.x < 64) {
    rank[threadIdx.x] = 0;
  }
  blockSync();
  const uint32_t mineAt = blockIdx.x * 64 + (threadIdx.x & 63);
  if (mineAt < n) {
    const uint64_t mine = keys[mineAt];
    const uint32_t segment = threadIdx.x >> 6;
    const uint32_t per = (n + 15) / 16;
    const uint32_t end = (segment + 1) * per < n ? (segment + 1) * per : n;
    uint32_t smaller = 0;
    for (uint32_t o = segment * per; o < end; ++o) {
      smaller += keys[o] < mine ? 1u : 0u;
    }
    atomicAdd(&rank[threadIdx.x & 63], smaller);
  }
  blockSync();
  // (more live rows than the host counted groups is the broken invariant *found reports: such ranks
  // must not be written behind the capacityOut entries of the list - *found sits right there)
  if (threadIdx.x < 64 && mineAt < n && rank[threadIdx.x] < capacityOut) {
    orderOut[rank[threadIdx.x]] = vals[mineAt];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *found = count;
  }
}

// The same for a table too large for one workgroup to scan: k_collect has listed the live rows,
// *count of them (<= kSmallSortMax, the host knows the number of groups); one workgroup ranks them.
__global__ __launch_bounds__(1024) void k_rank_sort_small(const uint64_t* firstIn, const uint32_t* indexIn,
                                                           const uint32_t* count, uint32_t* orderOut, uint32_t capacityOut) {
  __shared__ uint64_t keys[kSmallSortMax];
  __shared__ uint32_t rank[64];
  const uint32_t n = *count < static_cast<uint32_t>(kSmallSortMax) ? *count : static_cast<uint32_t>(kSmallSortMax);
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    keys[i] = firstIn[i];
  }
  if (threadIdx.x < 64) {
    rank[threadIdx.x] = 0;
  }
  blockSync();
  // (as in k_collect_sort_small: 64 entries per workgroup, 16 segments of the list per entry)
  const uint32_t mineAt = blockIdx.x * 64 + (threadIdx.x & 63);
  if (mineAt < n) {
    const uint64_t mine = keys[mineAt];
    const uint32_t segment = threadIdx.x >> 6;
    const uint32_t per = (n + 15) / 16;
    const uint32_t end = (segment + 1) * per < n ? (segment + 1) * per : n;
    uint32_t smaller = 0;
    for (uint32_t o = segment * per; o < end; ++o) {
      smaller += keys[o] < mine ? 1u : 0u;
    }
    atomicAdd(&rank[threadIdx.x & 63], smaller);
  }
  blockSync();
  if (threadIdx.x < 64 && mineAt < n && rank[threadIdx.x] < capacityOut) {
    orderOut[rank[threadIdx.x]] = indexIn[mineAt];
  }
}

// ---- output ---------------------------------------------------------------------
struct OutKey {
  void* values;
  uint64_t* nulls;
  int32_t kind;
  int32_t keyIndex;
  KeyRange range;
  const uint64_t* store;  // generic hash mode: key images by group id
  int32_t storeWords;
  int32_t pad;
};
struct OutAgg {
  void* values;
  uint64_t* nulls;
  void* values2;     // avg in partial/intermediate steps: the BIGINT count column
  uint64_t* nulls2;
  int32_t aggKind;   // vx355_agg_kind
  int32_t inputType;
  int32_t mainOff;
  int32_t loOff;     // DOUBLE sum / avg: the 'lo' word (value = hi + lo); BIGINT sum: the high word of
                     // the 128-bit total; -1 otherwise
  int32_t seenOff;   // count of contributing rows; -1 = never null; 1 = the first-row word
                     // (group exists <=> some row contributed)
  int32_t finalOut;
  int32_t pad;
};
struct ExtractArgs {
  const uint64_t* table;
  int32_t stride;
  int32_t mode;
  const uint32_t* order;  // group row indexes in output order
  int64_t begin;
  int32_t count;
  int32_t numKeys;
  int32_t numAggs;
  int32_t global;  // no keys: the single group is row 0
  const uint64_t* nullStore;  // generic hash mode
  uint32_t* overflow;         // set when a sum(BIGINT) total does not fit int64 (Counters::overflow)
  // the number of groups the small sort found, on its way to the pinned mailbox (checked behind the
  // page's synchronisation): carried by this launch instead of a copy command of its own
  const uint32_t* checkSrc;
  uint32_t* checkDst;
  OutKey keys[kMaxKeys];
  OutAgg aggs[kMaxAccs];
};

__device__ inline void writeBit(uint64_t* words, int32_t pos, bool bit) {
  uint64_t m = ballot(bit);
  if (words && (lane() == 0)) {
    words[pos >> 6] = m;
  }
}

__device__ inline void storeTyped(void* values, int32_t kind, int32_t pos, int64_t iv, double dv,
                                  bool asInt) {
  switch (kind) {
    case VX355_TINYINT:
      static_cast<int8_t*>(values)[pos] = static_cast<int8_t>(iv);
      break;
    case VX355_SMALLINT:
      static_cast<int16_t*>(values)[pos] = static_cast<int16_t>(iv);
      break;
    case VX355_INTEGER:
      static_cast<int32_t*>(values)[pos] = static_cast<int32_t>(iv);
      break;
    case VX355_BIGINT:
      static_cast<int64_t*>(values)[pos] = iv;
      break;
    case VX355_REAL:
      static_cast<float*>(values)[pos] = static_cast<float>(dv);
      break;
    case VX355_DOUBLE:
      static_cast<double*>(values)[pos] = dv;
      break;
    default:
      break;
  }
}

// One lane per output row; a wave covers 64 consecutive rows so null bitmaps
// and bit-packed BOOLEAN values are assembled with ballots.
__global__ __launch_bounds__(256) void k_extract(ExtractArgs a) {
  const int32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos == 0 && a.checkSrc != nullptr) {
    __hip_atomic_store(a.checkDst, *a.checkSrc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (pos - static_cast<int32_t>(lane()) >= a.count) {
    return;  // whole wave past the end: its bitmap word lies outside ceil(count / 64) words
  }
  const bool active = pos < a.count;
  uint64_t gi = 0;
  if (active) {
    gi = a.global ? 0 : a.order[a.begin + pos];
  }
  const uint64_t* g = a.table + gi * a.stride;
  // Four-word group rows (key, first row, one DOUBLE sum: config 4) are read as two 16-byte loads, up
  // front: with 10^8 groups every row is a random line of HBM, and reading its key word here and its
  // accumulator words a few hundred instructions later fetched the line twice (27.6 GB for 10^8
  // sparse groups where the direct-index table, whose key is the row number, fetched 14.8).
  const bool rowInRegisters = a.stride == 4;
  RpU64x2 rowLo{0, 0}, rowHi{0, 0};
  if (rowInRegisters && active) {
    rowLo = reinterpret_cast<const RpU64x2*>(g)[0];
    rowHi = reinterpret_cast<const RpU64x2*>(g)[1];
  }
  auto word = [&](int x) -> uint64_t {
    if (!rowInRegisters) {
      return g[x];
    }
    return x == 0 ? rowLo.x : (x == 1 ? rowLo.y : (x == 2 ? rowHi.x : rowHi.y));
  };
  const uint64_t key = a.mode == MODE_NORMALIZED ? (active ? word(0) : 0) : gi;
  for (int k = 0; k < a.numKeys; ++k) {
    const OutKey& ok = a.keys[k];
    if (ok.values == nullptr) {
      continue;  // the caller has the keys from elsewhere (DistinctPart::outer)
    }
    if (a.mode == MODE_HASH) {
      // Stored key images by group id.
      const bool valid = active && !((a.nullStore[gi] >> ok.keyIndex) & 1);
      writeBit(ok.nulls, pos, valid);
      const uint64_t w0 = active ? ok.store[gi * ok.storeWords] : 0;
      if (ok.kind == VX355_BOOLEAN) {
        writeBit(static_cast<uint64_t*>(ok.values), pos, valid && w0 != 0);
        continue;
      }
      if (!active) {
        continue;
      }
      if (ok.storeWords == 2) {
        const uint64_t w1 = ok.store[gi * 2 + 1];
        uint4 raw;
        raw.x = valid ? static_cast<uint32_t>(w0) : 0;
        raw.y = valid ? static_cast<uint32_t>(w0 >> 32) : 0;
        raw.z = valid ? static_cast<uint32_t>(w1) : 0;
        raw.w = valid ? static_cast<uint32_t>(w1 >> 32) : 0;
        static_cast<uint4*>(ok.values)[pos] = raw;
      } else if (ok.kind == VX355_REAL) {
        static_cast<float*>(ok.values)[pos] = valid ? __uint_as_float(static_cast<uint32_t>(w0)) : 0.f;
      } else if (ok.kind == VX355_DOUBLE) {
        static_cast<double*>(ok.values)[pos] = valid ? __longlong_as_double(static_cast<long long>(w0)) : 0.0;
      } else {
        storeTyped(ok.values, ok.kind, pos, valid ? static_cast<int64_t>(w0) : 0, 0, true);
      }
      continue;
    }
    uint64_t id = active ? (key / ok.range.multiplier) % ok.range.rangeSize : 0;
    const bool valid = active && id != 0;
    writeBit(ok.nulls, pos, valid);
    if (ok.kind == VX355_BOOLEAN) {
      writeBit(static_cast<uint64_t*>(ok.values), pos, valid && id == 2);
      continue;
    }
    if (!active) {
      continue;
    }
    const int64_t v = valid ? static_cast<int64_t>(id - 1 + static_cast<uint64_t>(ok.range.min)) : 0;
    if (ok.kind == VX355_VARCHAR || ok.kind == VX355_VARBINARY) {
      // Inverse of stringAsNumber: the marker bit sits right above the bytes.
      uint32_t size = 0;
      uint64_t bytes = 0;
      if (valid && v != 0) {
        int top = 63 - __clzll(static_cast<long long>(v));
        size = static_cast<uint32_t>(top >> 3);
        bytes = static_cast<uint64_t>(v) - (1ULL << top);
      }
      uint4 raw;
      raw.x = size;
      raw.y = static_cast<uint32_t>(bytes);
      raw.z = static_cast<uint32_t>(bytes >> 32);
      raw.w = 0;
      static_cast<uint4*>(ok.values)[pos] = raw;
    } else {
      storeTyped(ok.values, ok.kind, pos, v, 0, true);
    }
  }
  for (int j = 0; j < a.numAggs; ++j) {
    const OutAgg& oa = a.aggs[j];
    const uint64_t mainWord = active ? word(oa.mainOff) : 0;
    uint64_t seen = (active && oa.seenOff >= 0) ? word(oa.seenOff) : 1;
    if (oa.seenOff == 1) {
      seen = seen != kNoRow ? 1 : 0;
    }
    const bool valid = active && seen != 0;
    const bool inInt = oa.inputType <= VX355_BIGINT;
    switch (oa.aggKind) {
      case VX355_AGG_COUNT:
      case VX355_AGG_COUNT_STAR:
        writeBit(oa.nulls, pos, active);
        if (active) {
          static_cast<int64_t*>(oa.values)[pos] = static_cast<int64_t>(mainWord);
        }
        break;
      case VX355_AGG_SUM:
        writeBit(oa.nulls, pos, valid);
        if (active) {
          if (inInt) {
            // the 128-bit total must fit int64: high word = sign extension of the low word
            if (valid && oa.loOff >= 0 &&
                static_cast<int64_t>(word(oa.loOff)) != (static_cast<int64_t>(mainWord) < 0 ? -1 : 0)) {
              *a.overflow = 1;
            }
            static_cast<int64_t*>(oa.values)[pos] = valid ? static_cast<int64_t>(mainWord) : 0;
          } else {
            double d = valid ? __longlong_as_double(static_cast<long long>(mainWord)) +
                    __longlong_as_double(static_cast<long long>(word(oa.loOff)))
                             : 0.0;
            if (oa.inputType == VX355_REAL && oa.finalOut) {
              static_cast<float*>(oa.values)[pos] = static_cast<float>(d);
            } else {
              static_cast<double*>(oa.values)[pos] = d;
            }
          }
        }
        break;
      case VX355_AGG_MIN:
      case VX355_AGG_MAX:
        writeBit(oa.nulls, pos, valid);
        if (oa.inputType == VX355_BOOLEAN) {
          writeBit(static_cast<uint64_t*>(oa.values), pos, valid && orderedToInt64(mainWord) != 0);
        } else if (active) {
          if (inInt) {
            storeTyped(oa.values, oa.inputType, pos, valid ? orderedToInt64(mainWord) : 0, 0, true);
          } else {
            storeTyped(oa.values, oa.inputType, pos, 0, valid ? orderedToDouble(mainWord) : 0.0, false);
          }
        }
        break;
      default: {  // AVG
        writeBit(oa.nulls, pos, valid);
        const double sum = __longlong_as_double(static_cast<long long>(mainWord)) +
            (active ? __longlong_as_double(static_cast<long long>(word(oa.loOff))) : 0.0);
        const int64_t cnt = static_cast<int64_t>(seen);
        if (oa.finalOut) {
          if (active) {
            double v = valid ? sum / static_cast<double>(cnt) : 0.0;
            if (oa.inputType == VX355_REAL) {
              static_cast<float*>(oa.values)[pos] = static_cast<float>(v);
            } else {
              static_cast<double*>(oa.values)[pos] = v;
            }
          }
        } else {
          writeBit(oa.nulls2, pos, valid);
          if (active) {
            static_cast<double*>(oa.values)[pos] = valid ? sum : 0.0;
            static_cast<int64_t*>(oa.values2)[pos] = valid ? cnt : 0;
          }
        }
        break;
      }
    }
  }
}

// ---- GroupingSet::toIntermediate (GroupingSet.cpp:1589-1675) ---------------------------------
struct ToIntermediateAgg {
  ColView in, mask;
  int32_t aggKind;    // vx355_agg_kind
  int32_t hasIn, hasMask;
  int32_t inIsInt;
  int32_t inputType;
  int32_t pad;
  void* values;
  uint64_t* nulls;
  void* values2;      // avg: the BIGINT count column
  uint64_t* nulls2;
};

struct ToIntermediateArgs {
  int32_t numAggs;
  int32_t count;
  ToIntermediateAgg aggs[kMaxAccs];
};
static_assert(sizeof(ToIntermediateArgs) <= 4096, "kernel arguments are limited to 4 KB");

// One lane per input row: the PARTIAL step's output row of a group that holds this row only.
__global__ __launch_bounds__(256) void k_to_intermediate(ToIntermediateArgs a) {
  const int32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos - static_cast<int32_t>(lane()) >= a.count) {
    return;
  }
  const bool inRange = pos < a.count;
  for (int j = 0; j < a.numAggs; ++j) {
    const ToIntermediateAgg& g = a.aggs[j];
    bool active = inRange;
    if (active && g.hasMask) {
      active = !colIsNull(g.mask, pos) && loadInt64(g.mask, colIndex(g.mask, pos)) != 0;
    }
    if (active && g.hasIn) {
      active = !colIsNull(g.in, pos);
    }
    const int64_t i = (active && g.hasIn) ? colIndex(g.in, pos) : 0;
    switch (g.aggKind) {
      case VX355_AGG_COUNT:
      case VX355_AGG_COUNT_STAR:
        writeBit(g.nulls, pos, inRange);
        if (inRange) {
          static_cast<int64_t*>(g.values)[pos] = active ? 1 : 0;
        }
        break;
      case VX355_AGG_SUM:
        writeBit(g.nulls, pos, active);
        if (inRange) {
          if (g.inIsInt) {
            static_cast<int64_t*>(g.values)[pos] = active ? loadInt64(g.in, i) : 0;
          } else {
            static_cast<double*>(g.values)[pos] = active ? loadDouble(g.in, i) : 0.0;
          }
        }
        break;
      case VX355_AGG_MIN:
      case VX355_AGG_MAX:
        writeBit(g.nulls, pos, active);
        if (g.inputType == VX355_BOOLEAN) {
          writeBit(static_cast<uint64_t*>(g.values), pos, active && loadInt64(g.in, i) != 0);
        } else if (inRange) {
          if (g.inIsInt) {
            storeTyped(g.values, g.inputType, pos, active ? loadInt64(g.in, i) : 0, 0, true);
          } else {
            storeTyped(g.values, g.inputType, pos, 0, active ? loadDouble(g.in, i) : 0.0, false);
          }
        }
        break;
      default:  // AVG: ROW(DOUBLE sum, BIGINT count)
        writeBit(g.nulls, pos, active);
        writeBit(g.nulls2, pos, active);
        if (inRange) {
          static_cast<double*>(g.values)[pos] = active ? loadDouble(g.in, i) : 0.0;
          static_cast<int64_t*>(g.values2)[pos] = active ? 1 : 0;
        }
        break;
    }
  }
}

// ---- host side ---------------------------------------------------------------------

bool rawInput(int32_t step) { return step == VX355_STEP_PARTIAL || step == VX355_STEP_SINGLE; }
bool finalOutput(int32_t step) { return step == VX355_STEP_FINAL || step == VX355_STEP_SINGLE; }

int64_t typeMin(int32_t kind) {
  switch (kind) {
    case VX355_TINYINT:
      return INT8_MIN;
    case VX355_SMALLINT:
      return INT16_MIN;
    case VX355_INTEGER:
      return INT32_MIN;
    default:
      return INT64_MIN;
  }
}
int64_t typeMax(int32_t kind) {
  switch (kind) {
    case VX355_TINYINT:
      return INT8_MAX;
    case VX355_SMALLINT:
      return INT16_MAX;
    case VX355_INTEGER:
      return INT32_MAX;
    default:
      return INT64_MAX;
  }
}

struct PhysAcc {
  int32_t kind;
  int32_t inputCol;
  int32_t inputCol2 = -1;  // unused
  int32_t maskCol;
  bool inIsInt;
  int32_t aliasOf = -1;  // COUNT(col) == COUNT(*) while no batch had nulls in col
  bool valueNeeded = false;  // the count itself is an output (count / avg), not just a "seen" flag
  bool isLo = false;         // second word of a DOUBLE sum (exact remainder of the grid split)
  double splitM = 0;         // DOUBLE sum: 1.5 * 2^(G+52) of the grid, 0 = no split
};

struct LogicalAgg {
  vx355_agg_fn fn;
  int32_t main = -1;
  int32_t seen = -1;
};

struct KeyState {
  int32_t col;
  int32_t kind;
  bool hasObserved = false;
  int64_t obsMin = 0, obsMax = 0;
  KeyRange range;  // current device mapping (valid when tableReady)
};

constexpr int64_t kMaxRangeSpan = (1LL << 59) - 1;  // exec/VectorHasher.h:139 kMaxRange
constexpr size_t kCountersTemplateAt = 512;  // offset of the pristine Counters inside vx355_agg::countersBuf
constexpr size_t kCountersTicketAt = 1024;   // ... and of the ticket of publishCountersFromLastBlock
static_assert(sizeof(Counters) <= 512, "live counters, pristine copy and ticket sit 512 bytes apart");

}  // namespace
}  // namespace vx

using namespace vx;

struct vx355_agg {
  vx::Runtime* ctx = nullptr;  // this operator's execution context (stream, mailbox)
  vx::AsyncQueue* aq = nullptr;  // worker of vx355_agg_add_input_async (created on first use)
  // pages of vx355_agg_get_output_async by ticket, until vx355_agg_output_result hands them out
  struct QueuedPage {
    std::vector<vx355_out_column> cols;
    int32_t numRows = 0, finished = 0;
    int status = VX355_OK;
    std::string errorText;
    std::atomic<bool> complete{false};
  };
  std::mutex pagesMutex;
  std::map<int64_t, std::shared_ptr<QueuedPage>> pages;
  // vx355_agg_table_bytes: what get_stats would report, as of the last batch fed (written by
  // whichever thread feeds - the Driver thread or the queue's worker -, read without waiting)
  std::atomic<int64_t> publishedTableBytes{0};
  std::atomic<int64_t> publishedUsedBytes{0};  // vx355_agg_bytes_in_use
  std::atomic<int64_t> publishedGroups{0};
  int32_t step;
  bool ignoreNullKeys;
  std::vector<KeyState> keys;
  std::vector<LogicalAgg> aggs;
  std::vector<PhysAcc> phys;
  // Word of the group row each accumulator owns (row word = 2 + wordOf), -1 = none: a count that
  // only ever serves as a "group seen" flag is read off the first-row word, and a count(x) that
  // aliases count(*) gets a word the day a batch brings nulls in x (growWord re-lays the table).
  // BASELINE config 4 (sum(DOUBLE) per key): 4 words per group instead of 6.
  std::vector<int32_t> wordOf;
  std::vector<int32_t> outTypes;
  std::vector<int32_t> usedCols;
  std::vector<vx355_filter_term> fusedTerms;
  std::vector<vx355_projection> fusedProj;
  int32_t maxProjRef = -1;

  HostCoalescer coalescer;  // small host batches -> large launches

  // generic hash mode (keys without a normalized form)
  bool generic = false;
  DevBuf gSlots, gNullStore, gHashStore, gCounter;
  std::vector<DevBuf> gKeyStore;
  uint64_t gSlotCap = 0;
  uint64_t gMaxGroups = 0;
  // arena of grouping strings longer than 12 bytes: blocks are never moved (key images point into them)
  std::vector<DevBuf> strBlocks;
  DevBuf strCursor;          // u64: bytes used of the newest block
  uint64_t strCap = 0;       // capacity of the newest block
  uint64_t strUsedHost = 0;  // upper bound of what the kernels have taken from it
  bool hasStringKeys = false;
  std::vector<std::vector<char>> hostStrings;  // long strings of the last page handed to a host caller

  // device state
  int32_t mode = MODE_ARRAY;
  bool tableReady = false;
  DevBuf table;
  uint64_t capacity = 0;  // group rows
  int32_t stride = 0;
  DevBuf pattern;
  DevBuf countersBuf;
  DevBuf deferredBuf;
  DevBuf scratch, sortTmp;
  DevBuf ldsScratch;            // per-workgroup copies of the scratch flush (ldsGrid)
  bool ldsScratchFlush = true;  // VX355_AGG_SCRATCH_FLUSH=0: every LDS flush goes through HBM atomics
  int64_t scratchFlushes = 0;
  int scratchBlocksPerCu = 2;  // VX355_AGG_SCRATCH_BLOCKS_PER_CU: workgroups (= copies) per CU of a scratch-flush launch
  int64_t scratchMinAtomics = 256 << 10;  // VX355_AGG_SCRATCH_MIN_ATOMICS: flushes below this many HBM atomics keep them
  // radix-partitioned path (high cardinality)
  DevBuf rpRecs1, rpRecs2, rpHist, rpOffsets, rpTiles, rpMisc, rpScan, rpLayout2, rpLayout1, rpSplit;
  bool radixOptimistic = true;  // VX355_AGG_RADIX_OPTIMISTIC=0: level 2 always counts first
  bool radixOptimistic1 = true;  // VX355_AGG_RADIX_OPTIMISTIC1=0: level 1 always counts first (k_rp_count1)
  bool compactRecords = true;    // VX355_AGG_COMPACT_RECORDS=0: 16-byte records also when no group order is wanted (see recLoad)
  int64_t radixRedone = 0;      // level-2 passes redone exactly after a region overflowed
  int32_t hashSlotsFixed = 0;   // VX355_AGG_HASH_SLOTS: LDS entries of the hashed folds (0 = from a sample of the keys)
  bool foldGroups = true;       // VX355_AGG_FOLD_GROUPS=0: one partition per flush whatever the sample says
  bool coarseLevel2 = true;     // VX355_AGG_COARSE_LEVEL2=0: level 2 keeps the fan-out chosen before level 1
  int64_t coarsenedLaunches = 0;
  DevBuf rpSampleSets;          // k_rp_bucket_sample: a key set per sampled bucket + its counters
  int32_t lastHashSlots = 0;    // what the last hashed launch used
  int64_t radixMinRows = 4 << 20;
  int32_t radixMaxBins = kRadixMaxBins;  // widest single-level fan-out
  int64_t radixTileRows = 0;  // 0 = automatic
  int64_t radixLaunches = 0;
  int64_t compactLaunches = 0;  // of those, with 12-byte records (launchRadix, 'cr')
  bool radixSorted = true;   // VX355_AGG_RADIX_SORTED=0: scatter passes store record by record
  bool radixSparse = true;   // VX355_AGG_RADIX_SPARSE=0: open-addressing tables stay on k_agg_global
  // Open-addressing mode, operator without groups, a large batch: the radix folds append their
  // groups to a plain array of group rows (hashFoldFlushDense) and that array IS the table until
  // someone has to look a key up (tableDense: capacity = number of rows, all live; rebuildTable
  // re-keys it into a real open-addressing table before the next input is aggregated).
  bool radixDense = true;    // VX355_AGG_RADIX_DENSE=0: off
  bool tableDense = false;
  bool denseNext = false;    // the next launchChunk is such a launch
  int64_t denseMinRows = 1 << 22;   // VX355_AGG_DENSE_MIN_ROWS
  int64_t denseLaunches = 0, denseRefolds = 0, denseMerges = 0;
  DevBuf denseFlags;
  DevBuf cardSet;            // k_card_sample: [0] count, [16...] the set
  PinnedBuf outStage;        // small output pages leave through one pinned copy (getOutput)
  // The table was allocated but never written (rebuildTable skipped k_init_table because a radix
  // fold may come first and store every row itself); settleTable initialises it for anyone else.
  bool tableVirgin = false;
  // (first row, group row) pairs of the groups created so far, in orderKeys / orderVals: complete
  // while every group came out of an exclusive radix fold of the current table.
  int64_t pairCount = 0;
  bool pairsComplete = true;
  int64_t pairHoles = 0;   // entries of the list no group owns (dense folds: they sort behind every group)
  DevBuf orderKeys, orderVals, orderKeys2, orderVals2;
  const uint32_t* order = nullptr;
  int64_t numOutput = -1;  // set by finalize
  int64_t collectCheck = -1;  // groups k_collect_sort_small must have found (its count sits behind 'order')
  int64_t outputCursor = 0;
  bool noMoreInput = false;
  bool unorderedOutput = false;  // VX355_AGG_UNORDERED_OUTPUT: no first-seen sort
  bool flushing = false;   // vx355_agg_flush: the groups are being drained before noMoreInput
  int64_t numFlushes = 0;

  int64_t inputRows = 0;
  int64_t deferredRows = 0;
  int64_t numGroups = 0;
  int64_t numRehashes = 0;
  uint64_t arrayMax = 1ULL << 28;
  int64_t chunkRows = 1LL << 31;
  bool disableFast = false;
  int64_t jitLaunches = 0;
  bool jitEnabled = true;
  bool jitAsync = true;    // VX355_JIT=sync (or 1): a new plan shape waits ~0.8 s for hiprtc instead
  bool exactSums = true;
  bool sumGridsChosen = false;
  // Direct-index tables beyond the LDS path's one-entry-per-key map (capacity > 8192): how many
  // distinct keys a strided sample of the first batch holds (sampleCardinality). Few -> the LDS
  // kernels run with a hashed key -> slot map instead of handing every row to HBM atomics.
  bool cardSampled = false;   // (VX355_AGG_LDS_HASHED=0 sets it up front: no sample, no hashed map)
  int64_t sampledGroups = -1;
  int64_t firstRowsDistinct = 0;  // distinct keys among the first 2048 rows of the first batch (k_key_stats)
  // The sample found few keys (it did not saturate) in a key range far wider than the LDS kernels'
  // direct map: the operator keeps an open-addressing table instead of a direct-index one whose
  // rows would be 99.99 % empty - no multi-GB table to initialise and to scan for live rows, and the
  // radix path partitions by hash (rows spread evenly) instead of by key range (all rows of a key
  // in one partition: 124 ms for 200 M rows over 1800 keys against 17 ms).
  bool preferNormalized = false;
  bool slotsOnlyTables = true;   // VX355_AGG_SLOTS_ONLY=0: open-addressing tables always sized for a chunk of new groups
  bool slotsOnlyLaunch = false;  // the next launchChunk defers the rows of workgroups that run out of LDS slots
  bool logShapes = false;
  bool shapeLogged = false;
  int64_t deferCap = 1 << 20;
  int fastUnroll = 4;

  // agg(DISTINCT x), exec/DistinctAggregations.cpp: the SetAccumulator of one aggregate is a
  // second group table keyed on (grouping keys, x [, mask]) that sees every input batch; at
  // noMoreInput its rows, already in first-seen order, feed 'outer', an aggregation of x over
  // the grouping keys. 'outer' lists the same groups in the same order as this operator, so
  // get_output takes the aggregate's column from it row for row.
  //
  // min / max over VARCHAR / VARBINARY reuse the scheme (any step; the intermediate type is the
  // input type): dedup holds the distinct (keys, string [, mask]) rows, at noMoreInput the
  // strings are ranked on the device and 'outer' computes min / max over the BIGINT ranks; the
  // output column maps ranks back to string views (into dedup's arena for strings > 12 bytes).
  struct DistinctPart {
    int32_t specIndex = 0;  // position among the caller's aggregates
    vx355_agg* dedup = nullptr;
    vx355_agg* outer = nullptr;
    bool stringMinMax = false;
    std::vector<DevBuf> pairValues, pairNulls;  // dedup's rows, all at once (stringMinMax)
    DevBuf perm, permTmp, sortKeys, sortKeysTmp, rank, sortScratch, outRank, outRankNulls, outViews, outViewNulls;
    const uint32_t* sortedRows = nullptr;  // sorted position -> dedup row
  };
  std::vector<DistinctPart> distinct;
  std::vector<DistinctPart> retired;  // parts of the last flush: device output views point into their arenas
  // the caller's spec, kept to rebuild the parts after a partial flush
  std::vector<int32_t> specKeyCols, specKeyTypes;
  std::vector<vx355_agg_fn> specAggs;
  int32_t specFlags = 0;
  std::vector<int32_t> specIsDistinct;  // per caller aggregate
  std::vector<int32_t> specColBegin;    // first output column of caller aggregate i (+ one past the last)
  std::vector<int32_t> fullOutTypes;    // what the caller sees when 'distinct' is not empty
  bool keysOptional = false;  // an 'outer': key output columns without a buffer are skipped
  bool ownsCtx = true;        // dedup / outer run on their parent's context

  void dropParts(std::vector<DistinctPart>& parts) {
    for (auto& d : parts) {
      delete d.dedup;
      delete d.outer;
    }
    parts.clear();
  }
  ~vx355_agg() {
    dropParts(distinct);
    dropParts(retired);
  }

  int ldsBlocksPerCuCap = 0;       // VX355_AGG_LDS_BLOCKS_PER_CU (0: what LDS allows, at most the caller's bound)
  bool countersPublished = false;  // the chunk's last launch publishes the counters itself (ldsReduce)
  uint64_t pristineAtLaunch = 0;  // the context's launch count right behind k_init_state (ensureBasics)
  Counters* counters() { return countersBuf.as<Counters>(); }
};

namespace vx {
namespace {

bool flagFromFirstRow(const vx355_agg& h, int32_t i);

// Row word of accumulator i; only for accumulators that own one.
int32_t offOf(const vx355_agg& h, int32_t i) {
  if (h.wordOf.at(i) < 0) {
    VX_THROW(VX355_EINTERNAL, "accumulator without a word in the group row");
  }
  return 2 + h.wordOf[i];
}

void assignWords(vx355_agg& h) {
  h.wordOf.assign(h.phys.size(), -1);
  int32_t next = 0;
  for (size_t i = 0; i < h.phys.size(); ++i) {
    const auto& p = h.phys[i];
    if (p.isLo) {
      h.wordOf[i] = h.wordOf[i - 1] < 0 ? -1 : next++;  // right behind its first word
      continue;
    }
    if (p.aliasOf >= 0 || flagFromFirstRow(h, static_cast<int32_t>(i))) {
      continue;
    }
    h.wordOf[i] = next++;
  }
  h.stride = 2 + next;
}

int32_t findOrAddPhys(vx355_agg& h, int32_t kind, int32_t col, int32_t mask, bool inIsInt) {
  for (size_t i = 0; i < h.phys.size(); ++i) {
    const auto& p = h.phys[i];
    if (!p.isLo && p.kind == kind && p.inputCol == col && p.maskCol == mask && p.inIsInt == inIsInt) {
      return static_cast<int32_t>(i);
    }
  }
  PhysAcc p;
  p.kind = kind;
  p.inputCol = col;
  p.maskCol = mask;
  p.inIsInt = inIsInt;
  h.phys.push_back(p);
  const int32_t index = static_cast<int32_t>(h.phys.size() - 1);
  if (accWords(kind) == 2) {
    // DOUBLE sum: the exact remainders; BIGINT sum: the high word of the 128-bit total
    PhysAcc lo;
    lo.kind = accSecondKind(kind);
    lo.inputCol = -1;
    lo.maskCol = -1;
    lo.inIsInt = kind == ACC_SUM_I64;
    lo.isLo = true;
    h.phys.push_back(lo);  // always the word right after the first
  }
  return index;
}

// COUNT of non-null 'col' rows; starts as an alias of the row count under the
// same mask and is materialised the first time a batch brings nulls in 'col'.
int32_t countAcc(vx355_agg& h, int32_t col, int32_t mask) {
  int32_t star = findOrAddPhys(h, ACC_COUNT, -1, mask, true);
  if (col < 0) {
    return star;
  }
  int32_t c = findOrAddPhys(h, ACC_COUNT, col, mask, true);
  if (c == static_cast<int32_t>(h.phys.size()) - 1 && h.phys[c].aliasOf == -1 && c != star) {
    h.phys[c].aliasOf = star;
  }
  return c;
}

void buildPlan(vx355_agg& h, const vx355_agg_spec& spec) {
  const bool raw = rawInput(spec.step);
  const bool fin = finalOutput(spec.step);
  for (int32_t k = 0; k < spec.num_keys; ++k) {
    KeyState ks;
    ks.col = spec.key_cols[k];
    ks.kind = spec.key_types[k];
    if (kindWidth(ks.kind) < 0) {
      VX_THROW(VX355_EUNSUPPORTED, "group-by key type " + std::to_string(ks.kind));
    }
    if (!(isIntLike(ks.kind) || isString(ks.kind))) {
      h.generic = true;  // REAL / DOUBLE / TIMESTAMP have no value ids (VectorHasher.h:338-357)
    }
    h.hasStringKeys = h.hasStringKeys || isString(ks.kind);
    h.keys.push_back(ks);
    h.outTypes.push_back(ks.kind);
    h.usedCols.push_back(ks.col);
  }
  for (int32_t i = 0; i < spec.num_aggs; ++i) {
    LogicalAgg la;
    la.fn = spec.aggs[i];
    const auto& f = la.fn;
    const bool isInt = isIntLike(f.input_type);
    // count(x) only looks at x's nulls: any column type (CountAggregate.cpp:27-147)
    const bool countsAnyType = f.kind == VX355_AGG_COUNT && raw && kindWidth(f.input_type) >= 0;
    if (!(isInt || f.input_type == VX355_REAL || f.input_type == VX355_DOUBLE || countsAnyType)) {
      VX_THROW(VX355_EUNSUPPORTED, "aggregate input type " + std::to_string(f.input_type));
    }
    if (f.kind != VX355_AGG_COUNT_STAR || !raw) {
      VX_CHECK_ARG(f.input_col >= 0, "aggregate needs an input column");
    }
    if (f.input_col >= VX355_PROJECTION_COL_BASE) {
      // Input = projection of the fused FilterProject: always DOUBLE.
      if (f.input_type != VX355_DOUBLE || !rawInput(spec.step) ||
          (f.kind == VX355_AGG_COUNT_STAR)) {
        VX_THROW(VX355_EUNSUPPORTED, "projection inputs feed raw DOUBLE aggregates only");
      }
      h.maxProjRef = std::max(h.maxProjRef, f.input_col - VX355_PROJECTION_COL_BASE);
    } else {
      h.usedCols.push_back(f.input_col);
    }
    h.usedCols.push_back(f.input_col2);
    h.usedCols.push_back(f.mask_col);
    switch (f.kind) {
      case VX355_AGG_SUM:
        la.main = findOrAddPhys(h, isInt ? ACC_SUM_I64 : ACC_SUM_F64, f.input_col, f.mask_col, isInt);
        la.seen = countAcc(h, f.input_col, f.mask_col);
        h.outTypes.push_back(isInt ? VX355_BIGINT
                                   : ((f.input_type == VX355_REAL && fin) ? VX355_REAL : VX355_DOUBLE));
        break;
      case VX355_AGG_COUNT:
      case VX355_AGG_COUNT_STAR:
        if (raw) {
          la.main = countAcc(h, f.kind == VX355_AGG_COUNT ? f.input_col : -1, f.mask_col);
          h.phys[la.main].valueNeeded = true;
        } else {
          la.main = findOrAddPhys(h, ACC_SUM_I64_WRAP, f.input_col, f.mask_col, true);
        }
        h.outTypes.push_back(VX355_BIGINT);
        break;
      case VX355_AGG_MIN:
      case VX355_AGG_MAX:
        la.main = findOrAddPhys(h, f.kind == VX355_AGG_MIN ? ACC_MIN : ACC_MAX, f.input_col,
                                f.mask_col, isInt);
        la.seen = countAcc(h, f.input_col, f.mask_col);
        h.outTypes.push_back(f.input_type);
        break;
      case VX355_AGG_AVG:
        la.main = findOrAddPhys(h, ACC_SUM_F64, f.input_col, f.mask_col, false);
        if (raw) {
          la.seen = countAcc(h, f.input_col, f.mask_col);
          h.phys[la.seen].valueNeeded = true;
        } else {
          VX_CHECK_ARG(f.input_col2 >= 0, "avg over intermediate input needs the count column");
          // count = checkedPlus over the partial counts of rows whose sum is
          // not null (AverageAggregateBase.h:265-330). The accumulator is keyed
          // on the count column; rows are gated by the SUM column's nulls below.
          // (counts cannot leave int64: no 128-bit total for them)
          la.seen = findOrAddPhys(h, ACC_SUM_I64_WRAP, f.input_col2, f.mask_col, true);
        }
        if (fin) {
          h.outTypes.push_back(f.input_type == VX355_REAL ? VX355_REAL : VX355_DOUBLE);
        } else {
          h.outTypes.push_back(VX355_DOUBLE);
          h.outTypes.push_back(VX355_BIGINT);
        }
        break;
      default:
        VX_THROW(VX355_EUNSUPPORTED, "aggregate kind " + std::to_string(f.kind));
    }
    h.aggs.push_back(la);
  }
  size_t workAccs = 0;
  for (const auto& p : h.phys) {
    workAccs += p.isLo ? 0 : 1;
  }
  if (workAccs > static_cast<size_t>(kMaxAccs) || h.phys.size() > static_cast<size_t>(kMaxLdsAccs)) {
    VX_THROW(VX355_EUNSUPPORTED, "too many accumulators for one device table");
  }
  VX_CHECK_ARG(h.keys.size() <= static_cast<size_t>(kMaxKeys), "at most 8 grouping keys");
  assignWords(h);
}

// VectorHasher::extendRange (exec/VectorHasher.cpp:786-835) with the group-by
// reserve of 50 % (exec/HashTable.h:1213-1215).
void paddedRange(int32_t kind, int64_t mn, int64_t mx, int64_t* outMin, int64_t* outMax) {
  if (kind == VX355_BOOLEAN) {
    *outMin = 0;
    *outMax = 1;
    return;
  }
  const int64_t tMin = isString(kind) ? INT64_MIN : typeMin(kind);
  const int64_t tMax = isString(kind) ? INT64_MAX : typeMax(kind);
  const int64_t reserve = static_cast<int64_t>(2 + (mx - mn) * (50 / 100.0));
  *outMin = (tMin + reserve + 1 > mn) ? tMin : mn - reserve;
  *outMax = (tMax - reserve < mx) ? tMax : mx + reserve;
}

struct Decision {
  int32_t mode;
  uint64_t capacity;  // array mode: product of range sizes
  KeyRange ranges[kMaxKeys];
};

// The device analogue of HashTable::decideHashMode (HashTable.cpp:1751-1839),
// restricted to range-encoded keys: array while the product of the ranges fits
// the direct-index budget, else normalized key, else unsupported (kHash).
Decision decide(vx355_agg& h) {
  Decision d{};
  unsigned __int128 product = 1;
  bool overflow = false;
  // A single integer key is its own 64-bit normalized key whatever its range
  // (the reference falls back to kHash above kMaxRange; here the open-addressing
  // mode on id = value - INT64_MIN + 1 is the cheaper equivalent). Two ids are
  // reserved (0 = null, ~0 = empty slot): the two largest int64 values force
  // the generic mode.
  if (h.keys.size() == 1 && h.keys[0].kind >= VX355_TINYINT && h.keys[0].kind <= VX355_BIGINT &&
      h.keys[0].hasObserved) {
    const auto& ks = h.keys[0];
    int64_t span;
    if (__builtin_sub_overflow(ks.obsMax, ks.obsMin, &span) || span >= kMaxRangeSpan) {
      if (ks.obsMax > INT64_MAX - 2) {
        VX_THROW(VX355_EUNSUPPORTED, "grouping key uses the reserved ids of the wide single-key mode");
      }
      d.ranges[0].min = INT64_MIN;
      d.ranges[0].max = INT64_MAX - 2;
      d.ranges[0].rangeSize = ~0ULL;  // max - min + 2
      d.ranges[0].multiplier = 1;
      d.capacity = ~0ULL;
      d.mode = MODE_NORMALIZED;
      return d;
    }
  }
  for (size_t k = 0; k < h.keys.size(); ++k) {
    auto& ks = h.keys[k];
    int64_t mn = 0, mx = -1;
    uint64_t rangeSize;
    if (ks.kind == VX355_BOOLEAN) {
      mn = 0;
      mx = 1;
      rangeSize = 3;
    } else if (!ks.hasObserved) {
      // Only nulls so far: an empty range (every value will be deferred).
      mn = 0;
      mx = -1;
      rangeSize = 1;
    } else {
      int64_t span;
      if (__builtin_sub_overflow(ks.obsMax, ks.obsMin, &span) || span >= kMaxRangeSpan) {
        overflow = true;
        rangeSize = 1;
      } else {
        paddedRange(ks.kind, ks.obsMin, ks.obsMax, &mn, &mx);
        rangeSize = static_cast<uint64_t>(mx - mn) + 2;
      }
    }
    d.ranges[k].min = mn;
    d.ranges[k].max = mx;
    d.ranges[k].rangeSize = rangeSize;
    d.ranges[k].multiplier = static_cast<uint64_t>(product);
    product *= rangeSize;
    if (product >= (static_cast<unsigned __int128>(1) << 64) - 1) {
      overflow = true;
    }
  }
  if (overflow) {
    VX_THROW(VX355_EUNSUPPORTED,
             "grouping keys do not fit a 64-bit normalized key (generic hash mode not on device)");
  }
  d.capacity = static_cast<uint64_t>(product);
  // A global aggregation (no keys) is the one-row array table whatever the budget says.
  d.mode = (h.keys.empty() || d.capacity <= h.arrayMax) ? MODE_ARRAY : MODE_NORMALIZED;
  if (h.preferNormalized && !h.keys.empty() && d.capacity > 8192) {
    d.mode = MODE_NORMALIZED;  // few keys over a wide range: see sampleCardinality
  }
  return d;
}

void initTable(vx355_agg& h, DevBuf& buf, uint64_t rows) {
  buf.ensure(static_cast<size_t>(rows) * h.stride * 8 + 64);
  int grid = streamGrid(static_cast<int64_t>(rows) * h.stride, 256, 4);
  VX_LAUNCH("k_init_table", k_init_table, grid, 256, 0, buf.as<uint64_t>(), rows, h.stride,
            h.pattern.as<uint64_t>());
}

// Anything but a radix fold needs initialised group rows.
void settleTable(vx355_agg& h) {
  if (h.tableVirgin) {
    initTable(h, h.table, h.capacity);
    h.tableVirgin = false;
  }
}

// The operator's small device state in ONE launch (round 5: three uploads - pattern, counter template,
// counter reset - were three copy kernels in front of every operator's first batch, 15 us of config 1's
// 150-us step): the pattern words arrive as a kernel argument.
struct InitStateArgs {
  uint64_t pattern[64];
  int32_t stride;
};
__global__ __launch_bounds__(64) void k_init_state(InitStateArgs a, uint64_t* pattern, Counters* pristine, Counters* live,
                                                   uint32_t* ticket) {
  const int t = threadIdx.x;
  if (t < a.stride) {
    pattern[t] = a.pattern[t];
  }
  if (t == 0) {
    *ticket = 0;  // publishCountersFromLastBlock
  }
  // sizeof(Counters) is a multiple of 8: every lane writes words of both copies
  constexpr int kWords = static_cast<int>(sizeof(Counters) / 8);
  static_assert(sizeof(Counters) % 8 == 0 && offsetof(Counters, keyMin) % 8 == 0, "Counters as words");
  constexpr int kMinAt = static_cast<int>(offsetof(Counters, keyMin) / 8);
  constexpr int kMaxAt = static_cast<int>(offsetof(Counters, keyMax) / 8);
  for (int w = t; w < kWords; w += 64) {
    uint64_t v = 0;
    if (w >= kMinAt && w < kMinAt + kMaxKeys) {
      v = static_cast<uint64_t>(INT64_MAX);
    } else if (w >= kMaxAt && w < kMaxAt + kMaxKeys) {
      v = static_cast<uint64_t>(INT64_MIN);
    }
    reinterpret_cast<uint64_t*>(pristine)[w] = v;
    reinterpret_cast<uint64_t*>(live)[w] = v;
  }
}

void ensureBasics(vx355_agg& h) {
  if (h.pattern.ptr()) {
    return;
  }
  static_assert(2 + kMaxLdsAccs <= 64, "the pattern travels as a kernel argument");
  InitStateArgs ia{};
  ia.pattern[0] = kEmpty;
  ia.pattern[1] = kNoRow;
  for (size_t i = 0; i < h.phys.size(); ++i) {
    if (h.wordOf[i] >= 0) {
      ia.pattern[2 + h.wordOf[i]] = accIdentity(h.phys[i].kind);
    }
  }
  ia.stride = h.stride;
  h.pattern.ensure(static_cast<size_t>(h.stride) * 8);
  // [0] the live counters, [kCountersTemplateAt] a pristine copy: a reset is one device-to-device
  // copy queued on the stream (a pageable host source made every reset a blocking staged copy)
  h.countersBuf.ensure(kCountersTicketAt + 64);
  VX_LAUNCH("k_init_state", k_init_state, 1, 64, 0, ia, h.pattern.as<uint64_t>(),
            reinterpret_cast<Counters*>(static_cast<char*>(h.countersBuf.ptr()) + kCountersTemplateAt), h.counters(),
            reinterpret_cast<uint32_t*>(static_cast<char*>(h.countersBuf.ptr()) + kCountersTicketAt));
  h.pristineAtLaunch = Runtime::get().launchCount;  // (the live copy too: a resetCounters before any other launch has nothing to do)
}

void resetCounters(vx355_agg& h) {
  if (h.pristineAtLaunch != 0 && h.pristineAtLaunch == Runtime::get().launchCount) {
    return;  // k_init_state wrote them and no kernel has been launched since
  }
  copyIn(h.counters(), static_cast<const char*>(h.countersBuf.ptr()) + kCountersTemplateAt, VX355_MEM_DEVICE,
         sizeof(Counters));
}

// Through the context's pinned mailbox (words 64...): a real DMA transfer behind the kernels of the
// stream instead of the driver's pageable-destination path.
Counters readCounters(vx355_agg& h) {
  auto& rt = Runtime::get();
  static_assert(64 * 8 + sizeof(Counters) <= Mailbox::kWords * 8, "the mailbox holds the counters");
  HIP_OK(hipMemcpyAsync(rt.mail.host + 64, h.counters(), sizeof(Counters), hipMemcpyDeviceToHost, rt.stream));
  rt.sync();
  Counters c;
  std::memcpy(&c, rt.mail.host + 64, sizeof(c));
  return c;
}

// The counters into the pinned mailbox AND back to their pristine values, one launch: what
// readCounters followed by the next launch's resetCounters did with two copy commands.
__global__ __launch_bounds__(64) void k_read_reset_counters(uint64_t* live, const uint64_t* pristine, uint64_t* mailbox) {
  constexpr int kWords = static_cast<int>(sizeof(Counters) / 8);
  static_assert(sizeof(Counters) % 8 == 0, "whole words");
  for (int i = threadIdx.x; i < kWords; i += 64) {
    __hip_atomic_store(mailbox + i, live[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    live[i] = pristine[i];
  }
}

Counters readAndResetCounters(vx355_agg& h) {
  auto& rt = Runtime::get();
  VX_LAUNCH("k_read_reset_counters", k_read_reset_counters, 1, 64, 0, reinterpret_cast<uint64_t*>(h.counters()),
            reinterpret_cast<const uint64_t*>(static_cast<const char*>(h.countersBuf.ptr()) + kCountersTemplateAt),
            rt.mail.dev + 64);
  h.pristineAtLaunch = rt.launchCount;  // resetCounters: nothing to do until the next launch
  rt.d2hBytes += sizeof(Counters);
  rt.sync();
  Counters c;
  std::memcpy(&c, rt.mail.host + 64, sizeof(c));
  return c;
}

// For a launch whose last workgroup publishes the counters itself (publishCountersFromLastBlock).
constexpr int kMaxPublishingBlocks = 256;  // one ticket word: beyond a few hundred workgroups the tickets cost more than a launch
CounterMail counterMail(vx355_agg& h) {
  auto& rt = Runtime::get();
  char* base = static_cast<char*>(h.countersBuf.ptr());
  return CounterMail{reinterpret_cast<uint64_t*>(base), reinterpret_cast<const uint64_t*>(base + kCountersTemplateAt),
                     rt.mail.dev + 64, reinterpret_cast<uint32_t*>(base + kCountersTicketAt)};
}

// ... and what the host does behind such a launch instead of readAndResetCounters.
Counters takePublishedCounters(vx355_agg& h) {
  auto& rt = Runtime::get();
  h.pristineAtLaunch = rt.launchCount;
  rt.d2hBytes += sizeof(Counters);
  rt.sync();
  Counters c;
  std::memcpy(&c, rt.mail.host + 64, sizeof(c));
  return c;
}

void mergeObserved(vx355_agg& h, const Counters& c) {
  for (size_t k = 0; k < h.keys.size(); ++k) {
    if (c.keyMin[k] <= c.keyMax[k]) {
      auto& ks = h.keys[k];
      if (!ks.hasObserved) {
        ks.obsMin = c.keyMin[k];
        ks.obsMax = c.keyMax[k];
        ks.hasObserved = true;
      } else {
        ks.obsMin = std::min(ks.obsMin, c.keyMin[k]);
        ks.obsMax = std::max(ks.obsMax, c.keyMax[k]);
      }
    }
  }
}

uint64_t hashCapacityFor(uint64_t groups) {
  // HashTable::newHashTableEntries (exec/HashTable.h:946-956): load <= 0.7.
  uint64_t cap = std::max<uint64_t>(2048, nextPow2(groups + groups / 2 + 1));
  while (groups > cap * 7 / 10) {
    cap <<= 1;
  }
  return cap;
}

// (Re)creates the table for the current observations, moving live groups.
void rebuildTable(vx355_agg& h, uint64_t extraGroups) {
  auto& rt = Runtime::get();
  Decision d = decide(h);
  uint64_t newCap = d.capacity;
  if (d.mode == MODE_NORMALIZED) {
    newCap = hashCapacityFor(static_cast<uint64_t>(h.numGroups) + extraGroups);
  }
  DevBuf fresh;
  // An empty direct-index table beyond the LDS path: leave it unwritten, the first launch is
  // most likely a radix fold that stores every row itself (else settleTable).
  // (The same for a small one when the LDS kernels may flush through scratch copies: k_lds_reduce
  // then stores every word of every row.)
  const bool virgin = d.mode == MODE_ARRAY && h.numGroups == 0 && !h.keys.empty() &&
      (newCap > 8192 ? h.radixMinRows >= 0 : h.ldsScratchFlush);
  if (virgin) {
    fresh.ensure(static_cast<size_t>(newCap) * h.stride * 8 + 64);
  } else {
    initTable(h, fresh, newCap);
  }
  // group rows move: the listed pairs would point at the old places
  h.pairsComplete = h.numGroups == 0;
  h.pairCount = 0;
  h.pairHoles = 0;
  if (h.tableReady && h.numGroups > 0) {
    settleTable(h);
    RekeyArgs ra{};
    ra.oldTable = h.table.as<uint64_t>();
    ra.oldRows = h.capacity;
    ra.oldMode = h.mode;
    ra.newTable = fresh.as<uint64_t>();
    ra.newCapacity = newCap;
    ra.newMode = d.mode;
    ra.stride = h.stride;
    ra.numKeys = static_cast<int32_t>(h.keys.size());
    for (size_t k = 0; k < h.keys.size(); ++k) {
      ra.oldRange[k] = h.keys[k].range;
      ra.newRange[k] = d.ranges[k];
    }
    ra.counters = h.counters();
    VX_LAUNCH("k_rekey", k_rekey, streamGrid(static_cast<int64_t>(h.capacity), 256), 256, 0, ra);
    ++h.numRehashes;
    rt.sync();  // (k_rekey reads the old table, which the assignment below releases)
  }
  h.table = std::move(fresh);
  h.tableVirgin = virgin;
  h.tableDense = false;
  h.capacity = newCap;
  h.mode = d.mode;
  for (size_t k = 0; k < h.keys.size(); ++k) {
    h.keys[k].range = d.ranges[k];
  }
  h.tableReady = true;
}

// Picks the LDS layout for a launch. Returns false when the group range is too
// large for the LDS path.
constexpr int64_t kLdsHashedMaxGroups = 2048;

bool chooseLds(const vx355_agg& h, int numAccs, LdsPlan* plan, size_t* ldsBytes) {
  if ((h.mode != MODE_ARRAY && h.mode != MODE_NORMALIZED) || numAccs == 0) {
    return false;
  }
  // (two workgroups of up to 72 KB share a CU's 160 KB; BASELINE config 1's direct layout - 2003
  // possible keys x 3 words + the first-row words - takes 64 KB)
  const size_t budget = 72 * 1024;
  // ... unless twice that buys a replica PER LANE (REP 64): with 32 replicas lanes l and l + 32 of a wave
  // hit the same LDS word and every atomic of a row takes two passes. One workgroup per CU streams as
  // fast as three (profiles/r06_q1_stream_ceiling.md #6), so a plan with many accumulator words is better
  // off with 64 replicas in up to 128 KB: TPC-H Q1 (16 slots x 11 words: 90 KB) 6.91 - 6.96 -> 6.55 - 6.66 ms
  // per query on the same box (round 6). VX355_AGG_LDS_FULL_REPLICAS=0 keeps the smaller layouts.
  static const bool fullReplicas =
      !(std::getenv("VX355_AGG_LDS_FULL_REPLICAS") && std::atoi(std::getenv("VX355_AGG_LDS_FULL_REPLICAS")) == 0);
  const size_t bigBudget = fullReplicas ? std::min<size_t>(128 * 1024, Runtime::get().ldsPerBlock) : 0;
  const size_t accBytes = static_cast<size_t>(numAccs) * 8;
  if (h.mode == MODE_NORMALIZED || h.capacity > 8192) {
    // Too many possible keys for a map entry each (or an open-addressing table). Few of them
    // occurring (sampled on the first batch, then counted): a hashed map over the keys that occur,
    // slots = twice the groups, the slots keep the full normalized key.
    const int64_t groups = std::max<int64_t>(h.numGroups, h.sampledGroups);
    if (h.sampledGroups < 0 || groups > kLdsHashedMaxGroups) {
      return false;
    }
    // (slots are handed out one by one, so a quarter of headroom over the estimate is plenty; the
    // map keeps a load below 1/4; a workgroup that still runs out defers its rows)
    const uint64_t S = nextPow2(std::max<uint64_t>(16, static_cast<uint64_t>(groups + groups / 4 + 8)));
    const uint64_t M = 4 * S;
    auto bytes = [&](int rep) {
      return (((M + 2 * S + 2) * 4 + 15) & ~static_cast<size_t>(15)) + S * 8 + S * accBytes * rep;
    };
    // up to 60 KB two workgroups share a CU; hundreds of groups with several accumulator words may
    // take one CU's worth (128 KB, one workgroup per CU): still far ahead of one HBM atomic per row
    int rep = 0;
    for (int r = 64; r >= 1; r >>= 1) {
      // (never beyond what a workgroup of this device may allocate: 160 KB on gfx950, 64 KB elsewhere)
      if (bytes(r) <= (r == 1 ? std::min<size_t>(128 * 1024, Runtime::get().ldsPerBlock) : budget)) {
        rep = r;
        break;
      }
    }
    if (rep == 0) {
      return false;
    }
    // (not the replica-per-lane rule of the other layouts: with the hashed map one workgroup per CU is what
    // hurts - four-key Q1 with 4 instead of 2 replicas in 128 KB: 8.22 instead of 7.17 ms, round 6)
    plan->direct = 2;
    plan->mapWords = static_cast<int32_t>(M);
    plan->S = static_cast<int32_t>(S);
    plan->REP = rep;
    plan->A = numAccs;
    plan->capacity = h.capacity;
    *ldsBytes = bytes(rep);
    return true;
  }
  const uint64_t R = h.capacity;
  auto bytesFor = [&](uint64_t S, int rep, bool direct) {
    size_t words = (direct ? 0 : R) + 2 * S + 2;
    return ((words * 4 + 15) & ~static_cast<size_t>(15)) + S * accBytes * rep;
  };
  // Direct layout: one slot per possible key.
  int repDirect = 0;
  for (int rep = 64; rep >= 1; rep >>= 1) {
    if (bytesFor(R, rep, true) <= budget) {
      repDirect = rep;
      break;
    }
  }
  if (repDirect < 64 && repDirect > 0 && bytesFor(R, 64, true) <= bigBudget) {
    repDirect = 64;
  }
  // Compact layout: slots only for keys that occur, sized from what has been seen - before the first
  // launch has counted the groups, from the distinct keys among the first 2048 rows (key statistics) with
  // four times the headroom: TPC-H Q1's first chunk (1 M rows, 4 groups) otherwise runs on the direct
  // layout with 2 replicas of 209 slots, every lane of a wave on the same handful of words (159 us for
  // rows the compact layout folds in ~20; a workgroup that does run out of slots updates the table itself).
  const uint64_t seenGroups = h.numGroups > 0
      ? 2 * static_cast<uint64_t>(h.numGroups)
      : (h.firstRowsDistinct > 0 && h.firstRowsDistinct < 256 ? 4 * static_cast<uint64_t>(h.firstRowsDistinct) : 0);
  uint64_t S = nextPow2(std::max<uint64_t>(16, seenGroups));
  S = std::min<uint64_t>(S, nextPow2(R));
  int repCompact = 0;
  if (seenGroups > 0) {
    for (int rep = 64; rep >= 1; rep >>= 1) {
      if (bytesFor(S, rep, false) <= budget) {
        repCompact = rep;
        break;
      }
    }
    if (repCompact < 64 && repCompact > 0 && bytesFor(S, 64, false) <= bigBudget) {
      repCompact = 64;
    }
  }
  if (repDirect == 0 && repCompact == 0) {
    // Unknown cardinality and a range too wide for direct slots: as many
    // compact slots as fit.
    uint64_t s = 16;
    while (bytesFor(s * 2, 1, false) <= budget && s * 2 <= nextPow2(R)) {
      s *= 2;
    }
    if (bytesFor(s, 1, false) > budget) {
      return false;
    }
    S = s;
    repCompact = 1;
  }
  if (repDirect >= repCompact) {
    plan->direct = 1;
    plan->S = static_cast<int32_t>(R);
    plan->REP = repDirect;
  } else {
    plan->direct = 0;
    plan->S = static_cast<int32_t>(S);
    plan->REP = repCompact;
  }
  plan->A = numAccs;
  plan->capacity = R;
  *ldsBytes = bytesFor(plan->S, plan->REP, plan->direct != 0);
  return true;
}

// Maps a launch onto the shape-specialised kernel: fills the runtime arguments
// and the shape signature; false = the plan is outside the fast class.
bool buildFastArgs(const AggArgs& c, const LdsPlan& plan, FastArgs* f, FastSignature* sig) {
  if (c.rowList || c.rescanOld || c.numKeys < 1 || c.numKeys > kFastKeys ||
      c.numTerms > kFastTerms || c.numAccs > kFastAccs) {
    return false;
  }
  *f = FastArgs{};
  *sig = FastSignature{};
  // Columns that are flat, or dictionary wrapped by ONE shared index vector (FilterProject's
  // output), with or without a null bitmap; returns -1 = not eligible, 1 = wrapped.
  auto usable = [&](const ColView& v) -> int {
    if (v.enc == VX355_FLAT) {
      return 0;
    }
    if (v.enc == VX355_DICTIONARY && (f->indices == nullptr || f->indices == v.indices)) {
      f->indices = v.indices;
      return 1;
    }
    return -1;
  };
  for (int k = 0; k < c.numKeys; ++k) {
    const ColView& v = c.keys[k].col;
    const int wrapped = usable(v);
    if (wrapped < 0) {
      return false;
    }
    sig->ind |= static_cast<uint32_t>(wrapped) << fastKeyBit(k);
    if (v.nulls) {
      sig->nul |= 1u << fastKeyBit(k);
      f->keyNulls[k] = v.nulls;
    }
    int kind;
    if (isString(v.kind)) {
      // the kernel decodes strings of up to three bytes (fastKeyValue); a range that reaches
      // beyond them means longer keys occur: every such row would only be deferred and replayed
      if (c.keys[k].range.max >= (1LL << 25)) {
        return false;
      }
      kind = FK_VIEW;
    } else if (v.kind == VX355_INTEGER) {
      kind = FK_I32;
    } else if (v.kind == VX355_BIGINT) {
      kind = FK_I64;
    } else {
      return false;
    }
    if (k < 2) {
      (k == 0 ? sig->k0 : sig->k1) = kind;
    } else {
      sig->kx = (sig->kx & ~(15u << (4 * (k - 2)))) | (static_cast<uint32_t>(kind) << (4 * (k - 2)));
    }
    f->keyPtr[k] = v.values;
    f->range[k] = c.keys[k].range;
  }
  for (int t = 0; t < c.numTerms; ++t) {
    const TermArg& ta = c.terms[t];
    const int wrapped = usable(ta.col);
    if (wrapped < 0) {
      return false;
    }
    sig->ind |= static_cast<uint32_t>(wrapped) << (2 + t);
    if (ta.col.nulls) {
      sig->nul |= 1u << (2 + t);
      f->termNulls[t] = ta.col.nulls;
    }
    int kind;
    if (ta.constKind == VX355_BIGINT && ta.col.kind == VX355_INTEGER) {
      // The kernel compares in 32 bits: a constant outside int32 would wrap.
      if (ta.i64 < INT32_MIN || ta.i64 > INT32_MAX) {
        return false;
      }
      kind = FK_I32;
    } else if (ta.constKind == VX355_BIGINT && ta.col.kind == VX355_BIGINT) {
      kind = FK_I64;
    } else if (ta.constKind == VX355_DOUBLE && ta.col.kind == VX355_DOUBLE) {
      kind = FK_F64;
    } else {
      return false;
    }
    (t == 0 ? sig->t0 : sig->t1) = kind;
    f->term[t].ptr = ta.col.values;
    f->term[t].cmp = ta.cmp;
    f->term[t].i64 = ta.i64;
    f->term[t].f64 = ta.f64;
  }
  auto loadSlot = [&](const ColView& v) -> int {
    // operands reach the arithmetic as doubles (loadDouble): DOUBLE, REAL, BIGINT, INTEGER
    uint32_t lk;
    switch (v.kind) {
      case VX355_DOUBLE:
        lk = FL_F64;
        break;
      case VX355_REAL:
        lk = FL_F32;
        break;
      case VX355_BIGINT:
        lk = FL_I64;
        break;
      case VX355_INTEGER:
        lk = FL_I32;
        break;
      default:
        return -1;
    }
    const int wrapped = usable(v);
    if (wrapped < 0) {
      return -1;
    }
    for (int j = 0; j < sig->numLoads; ++j) {
      if (f->loadPtr[j] == v.values && ((sig->ind >> (4 + j)) & 1) == static_cast<uint32_t>(wrapped)) {
        return j;
      }
    }
    if (sig->numLoads == kFastLoads) {
      return -1;
    }
    f->loadPtr[sig->numLoads] = v.values;
    sig->lk |= lk << (2 * sig->numLoads);
    sig->ind |= static_cast<uint32_t>(wrapped) << (4 + sig->numLoads);
    if (v.nulls) {
      sig->nul |= 1u << (4 + sig->numLoads);
      f->loadNulls[sig->numLoads] = v.nulls;
    }
    return sig->numLoads++;
  };
  for (int j = 0; j < c.numAccs; ++j) {
    const AccArg& ac = c.accs[j];
    if (ac.hasMask) {
      // one flat BOOLEAN column per masked accumulator
      if (ac.mask.enc != VX355_FLAT || ac.mask.kind != VX355_BOOLEAN) {
        return false;
      }
      sig->msk |= 1u << j;
      f->maskBits[j] = static_cast<const uint64_t*>(ac.mask.values);
      f->maskNulls[j] = ac.mask.nulls;
    }
    uint64_t desc;
    if (ac.kind == ACC_COUNT && !ac.hasIn && ac.inProj < 0) {
      desc = accDesc(0);
    } else if (ac.kind == ACC_COUNT && ac.hasIn && ac.inProj < 0) {
      // count(x): rows where the DOUBLE column x is not null (gated by the load's null bit)
      const int slot = loadSlot(ac.in);
      if (slot < 0) {
        return false;
      }
      desc = accDesc(0, slot);
    } else if (ac.kind == ACC_COUNT && ac.inProj >= 0) {
      // count(projection): rows where none of its factor columns is null
      const ProjectionArg& pa = c.proj[ac.inProj];
      if (pa.numFactors > kFastFactors) {
        return false;
      }
      int l[kFastFactors] = {15, 15, 15};
      for (int q = 0; q < pa.numFactors; ++q) {
        if (pa.factors[q].hasCol) {
          l[q] = loadSlot(pa.factors[q].col);
          if (l[q] < 0) {
            return false;
          }
        }
      }
      desc = accDesc(0, l[0], l[1], l[2]);
    } else if (ac.kind == ACC_SUM_I64 || ac.kind == ACC_MIN || ac.kind == ACC_MAX) {
      // one unscaled operand: checked BIGINT sum of an integer column; min / max on the
      // order-preserving image of an integer (inIsInt) or floating column
      if (!ac.hasIn || ac.inProj >= 0) {
        return false;
      }
      const bool intCol = ac.in.kind == VX355_BIGINT || ac.in.kind == VX355_INTEGER;
      const bool floatCol = ac.in.kind == VX355_DOUBLE || ac.in.kind == VX355_REAL;
      uint64_t op;
      if (ac.kind == ACC_SUM_I64) {
        if (!intCol) {
          return false;
        }
        op = FO_SUM_I64;
      } else if (ac.inIsInt ? !intCol : !floatCol) {
        return false;
      } else {
        op = ac.kind == ACC_MIN ? (ac.inIsInt ? FO_MIN_I : FO_MIN_F) : (ac.inIsInt ? FO_MAX_I : FO_MAX_F);
      }
      const int slot = loadSlot(ac.in);
      if (slot < 0) {
        return false;
      }
      f->scale[j][0] = 1.0;
      f->offset[j][0] = 0.0;
      desc = accDesc(1, slot);
      sig->ops |= op << (4 * j);
    } else if (ac.kind != ACC_SUM_F64) {
      return false;
    } else if (ac.inProj >= 0) {
      const ProjectionArg& pa = c.proj[ac.inProj];
      if (pa.numFactors > kFastFactors) {
        return false;
      }
      int l[kFastFactors] = {15, 15, 15};
      for (int q = 0; q < pa.numFactors; ++q) {
        f->scale[j][q] = pa.factors[q].scale;
        f->offset[j][q] = pa.factors[q].offset;
        if (pa.factors[q].hasCol) {
          l[q] = loadSlot(pa.factors[q].col);
          if (l[q] < 0) {
            return false;
          }
        }
      }
      desc = accDesc(pa.numFactors, l[0], l[1], l[2]);
    } else {
      if (!ac.hasIn) {
        return false;
      }
      const int slot = loadSlot(ac.in);
      if (slot < 0) {
        return false;
      }
      f->scale[j][0] = 1.0;
      f->offset[j][0] = 0.0;
      desc = accDesc(1, slot);
    }
    f->splitM[j] = ac.kind == ACC_SUM_F64 ? ac.splitM : 0.0;
    if (j < 4) {
      sig->accLo |= desc << (16 * j);
    } else if (j < 8) {
      sig->accHi |= desc << (16 * (j - 4));
    } else {
      sig->accEx |= desc << (16 * (j - 8));
    }
  }
  sig->numAccs = c.numAccs;
  f->ignoreNullKeys = c.ignoreNullKeys;
  f->numRows = c.numRows;
  f->deferred = c.deferred;
  f->deferCap = c.deferCap;
  f->plan = plan;
  return true;
}

// ---- hiprtc instantiation of aggFastBody<Shape> for shapes outside the table ----
// The plan shape becomes template arguments of the same kernel body the
// ahead-of-time table uses (agg_device.h), compiled once per shape and process
// (~1 s) and cached; Velox's Wave backend does the same with NVRTC
// (experimental/wave/common/Compile.cu). Anything that goes wrong (no hiprtc,
// headers not next to the library) disables the JIT and the generic kernel runs.
struct JitKernel {
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;
};

struct JitState {
  std::mutex mutex;
  std::map<std::string, JitKernel> kernels;
  std::map<std::string, std::shared_future<std::vector<char>>> pending;  // VX355_JIT=async: code objects on their way
  bool disabled = false;
  std::string csrcDir;
  std::string clangInclude;
  // Code objects on disk, shared by every process of the machine: <cacheDir>/<hash of the shape,
  // the architecture and the device headers>.hsaco. A new worker process loads an instance in
  // ~1 ms instead of compiling it for ~0.8 s (and, with the asynchronous compile, instead of running
  // its first seconds on the interpreting kernel). VX355_CACHE_DIR; "" or unwritable = no cache.
  std::string cacheDir;
  uint64_t sourceHash = 0;
  int64_t compiled = 0, loadedFromDisk = 0;
};

JitState& jitState() {
  static JitState st;
  return st;
}

bool jitPrepare(JitState& st) {
  if (!st.csrcDir.empty()) {
    return true;
  }
  Dl_info info;
  if (!dladdr(reinterpret_cast<const void*>(&vx355_agg_create), &info) || !info.dli_fname) {
    st.disabled = true;
    return false;
  }
  std::string lib = info.dli_fname;
  const size_t slash = lib.rfind('/');
  st.csrcDir = (slash == std::string::npos ? std::string(".") : lib.substr(0, slash)) + "/csrc";
  glob_t g{};
  if (glob("/opt/rocm/lib/llvm/lib/clang/*/include", 0, nullptr, &g) == 0 && g.gl_pathc > 0) {
    st.clangInclude = g.gl_pathv[g.gl_pathc - 1];
  }
  globfree(&g);
  if (FILE* f = fopen((st.csrcDir + "/agg_device.h").c_str(), "r")) {
    fclose(f);
  } else {
    st.disabled = true;
    return false;
  }
  // what an instance is compiled from: the three device headers (FNV-1a over their bytes)
  uint64_t h = 1469598103934665603ULL;
  for (const char* name : {"/agg_device.h", "/device_utils.h", "/expr_device.h"}) {
    if (FILE* f = fopen((st.csrcDir + name).c_str(), "rb")) {
      unsigned char buf[4096];
      size_t got;
      while ((got = fread(buf, 1, sizeof(buf), f)) > 0) {
        for (size_t i = 0; i < got; ++i) {
          h = (h ^ buf[i]) * 1099511628211ULL;
        }
      }
      fclose(f);
    }
  }
  st.sourceHash = h;
  if (const char* e = std::getenv("VX355_CACHE_DIR")) {
    st.cacheDir = e;
  } else if (const char* x = std::getenv("XDG_CACHE_HOME")) {
    st.cacheDir = std::string(x) + "/vx355";
  } else if (const char* home = std::getenv("HOME")) {
    st.cacheDir = std::string(home) + "/.cache/vx355";
  }
  if (!st.cacheDir.empty()) {
    // (mkdir -p of the last two components; failure just means no cache)
    const size_t slash = st.cacheDir.rfind('/');
    if (slash != std::string::npos && slash > 0) {
      (void)mkdir(st.cacheDir.substr(0, slash).c_str(), 0755);
    }
    (void)mkdir(st.cacheDir.c_str(), 0755);
  }
  return true;
}

std::string jitCachePath(const JitState& st, const std::string& key) {
  if (st.cacheDir.empty()) {
    return std::string();
  }
  uint64_t h = st.sourceHash;
  for (const char* part : {key.c_str(), "|gfx950|O3|ffp-contract=off"}) {
    for (const char* p = part; *p; ++p) {
      h = (h ^ static_cast<unsigned char>(*p)) * 1099511628211ULL;
    }
  }
  char name[64];
  snprintf(name, sizeof(name), "/agg_fast_%016llx.hsaco", static_cast<unsigned long long>(h));
  return st.cacheDir + name;
}

std::vector<char> jitReadCache(const std::string& path) {
  std::vector<char> code;
  if (path.empty()) {
    return code;
  }
  if (FILE* f = fopen(path.c_str(), "rb")) {
    if (fseek(f, 0, SEEK_END) == 0) {
      const long size = ftell(f);
      if (size > 0 && fseek(f, 0, SEEK_SET) == 0) {
        code.resize(static_cast<size_t>(size));
        if (fread(code.data(), 1, code.size(), f) != code.size()) {
          code.clear();
        }
      }
    }
    fclose(f);
  }
  return code;
}

// Atomically (write to a temporary name, rename): concurrent processes compiling the same shape
// cannot leave a torn file behind.
void jitWriteCache(const std::string& path, const std::vector<char>& code) {
  if (path.empty() || code.empty()) {
    return;
  }
  const std::string tmp = path + "." + std::to_string(static_cast<long long>(getpid())) + ".tmp";
  if (FILE* f = fopen(tmp.c_str(), "wb")) {
    const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
    fclose(f);
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) {
      (void)remove(tmp.c_str());
    }
  }
}

// A process that exits while a background compile is still inside hiprtc / comgr dies in LLVM's torn-down
// statics (seen as an intermittent SIGSEGV / abort "In function: k_agg_fast_jit" of a short-lived program).
// exit() therefore waits for the compiles in flight: the handler is registered both when a compile is
// launched (before comgr is loaded: it then runs after comgr's own exit handlers were registered, i.e. it
// holds exit() while the helper thread finishes) and from the helper thread once hiprtc has loaded comgr
// (registered later than comgr's statics, so it runs BEFORE they are destroyed).
// Round 6: that was not enough. LLVM constructs function-local statics DURING a compilation; they register
// their destructors after both handlers above, so exit() destroys them first and only then reaches the
// handler that waits - a program that finished while its first background compile was still running (every
// first run against an empty instance cache) aborted in LLVM ("Cannot implicitly convert a scalable size
// ..."), crashed or hung about one time in five (profiles/r06_jit_exit_race.md). Therefore the FIRST
// compilation of a process runs on the calling thread (gJitWarm: ~0.8 s once, and only when the on-disk
// cache misses): when it returns, the compiler's statics exist, the handler registered after it runs before
// their destructors, and every later background compile is covered.
std::atomic<int> gJitInFlight{0};
std::atomic<bool> gJitWarm{false};
void waitForJitAtExit() {
  for (int i = 0; i < 6000 && gJitInFlight.load(std::memory_order_acquire) > 0; ++i) {
    std::this_thread::sleep_for(std::chrono::milliseconds(10));
  }
}

// hiprtc half of an instantiation: CPU only, may run on any thread. Empty result = failure.
std::vector<char> jitCompile(const std::string& src, const std::string& clangInclude, std::string* buildLog) {
  // ONE compilation at a time in the process (hiprtc serialises its entry points itself; this keeps the
  // order of "first compile, then the exit handler" below well defined when several Drivers miss at once).
  static std::mutex oneCompiler;
  std::lock_guard<std::mutex> serial(oneCompiler);
  hiprtcProgram prog = nullptr;
  bool ok = hiprtcCreateProgram(&prog, src.c_str(), "vx355_agg_fast_jit.hip", 0, nullptr, nullptr) == HIPRTC_SUCCESS;
  static std::once_flag afterComgr;
  std::call_once(afterComgr, [] { std::atexit(waitForJitAtExit); });
  if (ok) {
    const std::string inc1 = "-I/opt/rocm/include";
    const std::string inc2 = "-I" + clangInclude;
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", inc1.c_str(),
                          inc2.c_str()};
    ok = hiprtcCompileProgram(prog, 6, opts) == HIPRTC_SUCCESS;
    size_t logSize = 0;
    if (hiprtcGetProgramLogSize(prog, &logSize) == HIPRTC_SUCCESS && logSize > 1) {
      buildLog->resize(logSize);
      hiprtcGetProgramLog(prog, &(*buildLog)[0]);
    }
  }
  std::vector<char> code;
  if (ok) {
    size_t size = 0;
    ok = hiprtcGetCodeSize(prog, &size) == HIPRTC_SUCCESS && size > 0;
    if (ok) {
      code.resize(size);
      ok = hiprtcGetCode(prog, code.data()) == HIPRTC_SUCCESS;
    }
  }
  if (prog) {
    hiprtcDestroyProgram(&prog);
  }
  if (!ok) {
    code.clear();
  }
  if (!gJitWarm.exchange(true)) {
    std::atexit(waitForJitAtExit);  // registered AFTER the compiler's lazily constructed statics: runs before their destructors
  }
  return code;
}

// async: the compilation runs on a helper thread and this call returns nullptr until the code
// object is there (the caller launches the interpreting kernel meanwhile): a new plan shape does
// not stall the Driver thread for the ~0.8 s hiprtc needs (VX355_JIT=async).
hipFunction_t jitFastKernel(const FastSignature& sig, int unroll, bool log, bool async) {
  JitState& st = jitState();
  std::lock_guard<std::mutex> lock(st.mutex);  // operators on several Driver threads share the cache
  if (st.disabled || !jitPrepare(st)) {
    return nullptr;
  }
  char key[256];
  snprintf(key, sizeof(key), "%d,%d,%d,%d,%d,%d,%d,%lluull,%lluull,%uu,%uu,%lluull,%uu,%uu,%lluull,%uu", unroll, sig.k0,
           sig.k1, sig.t0, sig.t1, sig.numLoads, sig.numAccs, static_cast<unsigned long long>(sig.accLo),
           static_cast<unsigned long long>(sig.accHi), sig.ind, sig.nul, static_cast<unsigned long long>(sig.accEx),
           sig.lk, sig.msk, static_cast<unsigned long long>(sig.ops), sig.kx);
  // A loaded module belongs to one GPU.
  const std::string cacheKey = std::to_string(Runtime::get().device) + ":" + key;
  auto it = st.kernels.find(cacheKey);
  if (it != st.kernels.end()) {
    return it->second.fn;
  }
  const std::string src = "#include \"" + st.csrcDir + "/agg_device.h\"\n"
      "using S = vx::FastShape<" + std::string(key) + ">;\n"
      "extern \"C\" __global__ __launch_bounds__(512, 4) void k_agg_fast_jit(vx::FastArgs a) {\n"
      "  vx::aggFastBody<S>(a);\n}\n";
  std::vector<char> code;
  std::string buildLog;
  const std::string cachePath = jitCachePath(st, key);
  auto pend = st.pending.find(key);
  if (pend != st.pending.end()) {
    if (pend->second.wait_for(std::chrono::seconds(0)) != std::future_status::ready) {
      return nullptr;  // still compiling
    }
    code = pend->second.get();
    ++st.compiled;
    jitWriteCache(cachePath, code);
  } else if (!(code = jitReadCache(cachePath)).empty()) {
    ++st.loadedFromDisk;
    if (log) {
      fprintf(stderr, "vx355: instance of shape %s loaded from %s\n", key, cachePath.c_str());
    }
  } else if (async && gJitWarm.load()) {
    const std::string inc = st.clangInclude;
    static std::once_flag beforeComgr;
    std::call_once(beforeComgr, [] { std::atexit(waitForJitAtExit); });
    gJitInFlight.fetch_add(1, std::memory_order_acq_rel);
    st.pending[key] = std::async(std::launch::async, [src, inc]() {
                        std::string ignored;
                        auto code = jitCompile(src, inc, &ignored);
                        gJitInFlight.fetch_sub(1, std::memory_order_acq_rel);
                        return code;
                      }).share();
    if (log) {
      fprintf(stderr, "vx355: compiling an instance of shape %s in the background\n", key);
    }
    return nullptr;
  } else {
    if (log) {
      fprintf(stderr, "vx355: compiling an instance of shape %s\n", key);
    }
    code = jitCompile(src, st.clangInclude, &buildLog);
    ++st.compiled;
    jitWriteCache(cachePath, code);
  }
  JitKernel k;
  bool ok = !code.empty();
  if (ok) {
    ok = hipModuleLoadData(&k.module, code.data()) == hipSuccess &&
        hipModuleGetFunction(&k.fn, k.module, "k_agg_fast_jit") == hipSuccess;
  }
  if (!ok) {
    (void)hipGetLastError();
    if (log) {
      fprintf(stderr, "vx355: hiprtc instantiation failed for shape %s\n%s\n", key, buildLog.c_str());
    }
    k = JitKernel{};
  }
  st.kernels[cacheKey] = k;  // failures are cached too: one attempt per shape
  return k.fn;
}

void launchJitFast(hipFunction_t fn, const FastArgs& fa, int grid, size_t ldsBytes) {
  auto& rt = Runtime::get();
  size_t size = sizeof(FastArgs);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<FastArgs*>(&fa),
                    HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  ++rt.launchCount;  // (as VX_LAUNCH does: resetCounters compares it)
  if (rt.profile) {
    rt.profBegin("k_agg_fast");
  }
  HIP_OK(hipModuleLaunchKernel(fn, grid, 1, 1, 512, 1, 1, static_cast<unsigned>(ldsBytes), rt.stream, nullptr,
                               config));
  if (rt.profile) {
    rt.profEnd("k_agg_fast");
  }
}

const FastEntry* findFastEntry(const FastSignature& sig, int unroll) {
  const FastEntry* any = nullptr;
  for (const auto& e : kFastTable) {
    if (e.sig == sig) {
      if (e.unroll == unroll) {
        return &e;
      }
      any = &e;
    }
  }
  return any;
}

// A count that only serves as the "seen" flag of sum/min/max over a column
// without nulls or mask equals "the group exists", which the first-row word
// already records: such a count(*) is not maintained at all.
bool flagFromFirstRow(const vx355_agg& h, int32_t i) {
  const auto& p = h.phys[i];
  if (p.kind != ACC_COUNT || p.aliasOf >= 0 || p.inputCol >= 0 || p.maskCol >= 0 || p.valueNeeded) {
    return false;
  }
  for (const auto& q : h.phys) {
    if (q.aliasOf == i && q.valueNeeded) {
      return false;
    }
  }
  return true;
}

void fillAccArgs(vx355_agg& h, const DeviceBatch& db, AggArgs* a) {
  int n = 0;
  for (size_t i = 0; i < h.phys.size(); ++i) {
    const auto& p = h.phys[i];
    if (p.isLo || p.aliasOf >= 0 || flagFromFirstRow(h, static_cast<int32_t>(i))) {
      continue;
    }
    AccArg& aa = a->accs[n++];
    aa = AccArg{};
    aa.splitM = p.splitM;
    aa.kind = p.kind;
    aa.inIsInt = p.inIsInt ? 1 : 0;
    aa.off = offOf(h, static_cast<int32_t>(i));
    aa.phys = static_cast<int32_t>(i);
    aa.inProj = -1;
    if (p.inputCol >= VX355_PROJECTION_COL_BASE) {
      aa.inProj = p.inputCol - VX355_PROJECTION_COL_BASE;
    } else if (p.inputCol >= 0) {
      aa.hasIn = 1;
      aa.in = db.col(p.inputCol);
    }
    if (p.maskCol >= 0) {
      aa.hasMask = 1;
      aa.mask = db.col(p.maskCol);
    }
  }
  a->numAccs = n;
}

// avg over intermediate input gates its count on the nulls of the SUM column.
void patchAvgIntermediate(vx355_agg& h, const DeviceBatch& db, AggArgs* a) {
  if (rawInput(h.step)) {
    return;
  }
  for (const auto& la : h.aggs) {
    if (la.fn.kind != VX355_AGG_AVG) {
      continue;
    }
    for (int j = 0; j < a->numAccs; ++j) {
      if (a->accs[j].phys == la.seen) {
        // Null sums exclude the row: fold the sum column's nulls into the mask
        // slot when the count column itself carries no nulls of its own.
        ColView sumCol = db.col(la.fn.input_col);
        if (sumCol.nulls && !a->accs[j].in.nulls) {
          a->accs[j].in.nulls = sumCol.nulls;
        }
      }
    }
  }
}

// Accumulator i gets a word of its own: the group rows grow by one word (new word = identity).
void growWord(vx355_agg& h, int32_t i) {
  if (h.wordOf[i] >= 0) {
    return;
  }
  auto& rt = Runtime::get();
  const int32_t oldStride = h.stride;
  h.wordOf[i] = oldStride - 2;
  h.stride = oldStride + 1;
  // the pattern follows the layout
  ensureBasics(h);
  std::vector<uint64_t> pat(h.stride);
  pat[0] = kEmpty;
  pat[1] = kNoRow;
  for (size_t q = 0; q < h.phys.size(); ++q) {
    if (h.wordOf[q] >= 0) {
      pat[2 + h.wordOf[q]] = accIdentity(h.phys[q].kind);
    }
  }
  h.pattern.ensure(pat.size() * 8 + 64);
  copyIn(h.pattern.ptr(), pat.data(), VX355_MEM_HOST, pat.size() * 8);
  rt.sync();
  if (!h.tableReady) {
    return;
  }
  DevBuf fresh;
  fresh.ensure(static_cast<size_t>(h.capacity) * h.stride * 8 + 64);
  if (!h.tableVirgin) {
    VX_LAUNCH("k_restride", k_restride, streamGrid(static_cast<int64_t>(h.capacity) * h.stride, 256, 4), 256, 0,
              h.table.as<uint64_t>(), fresh.as<uint64_t>(), h.capacity, oldStride, h.stride, h.pattern.as<uint64_t>());
    rt.sync();
  }
  h.table = std::move(fresh);
}

void materializeAliases(vx355_agg& h, const DeviceBatch& db) {
  for (size_t i = 0; i < h.phys.size(); ++i) {
    auto& p = h.phys[i];
    if (p.aliasOf < 0 || p.inputCol < 0) {
      continue;
    }
    bool mayBeNull = false;
    if (p.inputCol >= VX355_PROJECTION_COL_BASE) {
      const auto& pr = h.fusedProj.at(p.inputCol - VX355_PROJECTION_COL_BASE);
      for (int f = 0; f < pr.num_factors; ++f) {
        if (pr.factors[f].col >= 0 && db.col(pr.factors[f].col).nulls) {
          mayBeNull = true;
        }
      }
    } else {
      mayBeNull = db.col(p.inputCol).nulls != nullptr;
    }
    if (!mayBeNull) {
      continue;
    }
    const int32_t srcOff = flagFromFirstRow(h, p.aliasOf) ? 1 : offOf(h, p.aliasOf);
    growWord(h, static_cast<int32_t>(i));
    if (h.tableReady && !h.tableVirgin) {
      VX_LAUNCH("k_copy_acc", k_copy_acc, streamGrid(static_cast<int64_t>(h.capacity), 256), 256, 0,
                h.table.as<uint64_t>(), h.capacity, h.stride, srcOff, offOf(h, static_cast<int32_t>(i)));
    }
    p.aliasOf = -1;
  }
}

int log2Ceil(uint64_t v) {
  int l = 0;
  while ((1ULL << l) < v) {
    ++l;
  }
  return l;
}

// Groups per partition of the radix path: B x (accumulators x 8 + 4) bytes of
// LDS per workgroup, two workgroups per CU.
int radixShiftB(int numWords) { return numWords <= 1 ? 12 : (numWords <= 4 ? 11 : 10); }

// LDS words per group of the fold: a DOUBLE sum owns two (hi, lo).
int radixWords(const AggArgs& a) {
  int n = 0;
  for (int j = 0; j < a.numAccs; ++j) {
    n += accWords(a.accs[j].kind);
  }
  return n;
}

// Open-addressing tables take the radix path too (partitioned by home slot, k_rp_aggregate_hashed):
// records of at most 4 words = word 0 + up to two operands + the full key; the sorted scatters only.
// Home slots per partition of a hashed launch: as many partitions as give ~kHashRecsPerPart
// records each (a power of two, at most 2^20 = two levels of 1024 bins, at least 2).
int radixHashedShift(uint64_t capacity, int64_t rows) {
  const int capBits = log2Ceil(capacity);
  int partBits = log2Ceil(static_cast<uint64_t>(std::max<int64_t>(2, rows / kHashRecsPerPart)));
  partBits = std::max(1, std::min({partBits, 20, capBits - 4}));
  return capBits - partBits;
}

// Slot space of a dense launch over 'rows' rows: partitions of ~kHashRecsPerPart records (at most
// 2^20 of them), kHashSlots virtual slots each.
uint64_t denseSlotSpace(int64_t rows) {
  const int partBits =
      std::max(1, std::min(20, log2Ceil(static_cast<uint64_t>(std::max<int64_t>(2, rows / kHashRecsPerPart)))));
  return 1ULL << (partBits + log2Ceil(static_cast<uint64_t>(kHashSlots)));
}

int radixHashedVals(const AggArgs& a) {
  int numVals = 0;
  for (int j = 0; j < a.numAccs; ++j) {
    numVals += a.accs[j].kind == ACC_COUNT ? 0 : 1;
  }
  return numVals;
}

// A batch into an operator in open-addressing mode that has no groups yet (see tableDense).
bool radixDenseEligible(const vx355_agg& h, const AggArgs& a, int64_t rows) {
  return h.radixDense && h.radixSparse && h.radixSorted && h.radixMinRows >= 0 && h.mode == MODE_NORMALIZED &&
      h.numGroups == 0 && a.numAccs >= 1 && a.numAccs <= kRadixMaxAccs && radixHashedVals(a) <= 2 &&
      rows >= std::max<int64_t>(h.radixMinRows, h.denseMinRows);
}

// Rows one dense launch takes: row numbers must fit the record, and the two record buffers stay
// under ~30 % of the GPU's memory.
int64_t denseMaxRows(const AggArgs& a) {
  size_t freeBytes = 0, totalBytes = 0;
  HIP_OK(hipMemGetInfo(&freeBytes, &totalBytes));
  const int64_t recBytes = (2 + radixHashedVals(a)) * 8;
  const int64_t byMemory = static_cast<int64_t>(totalBytes * 3 / 10) / (2 * recBytes);
  int64_t rows = std::max<int64_t>(1 << 22, std::min<int64_t>(1LL << radixRowBits(denseSlotSpace(1LL << 30)), byMemory));
  if (const char* e = std::getenv("VX355_AGG_DENSE_MAX_ROWS")) {
    rows = std::max<int64_t>(64, std::strtoll(e, nullptr, 10));
  }
  return rows & ~63LL;
}

bool radixHashedEligible(const vx355_agg& h, const AggArgs& a) {
  if (a.mode != MODE_NORMALIZED || !h.radixSparse || !h.radixSorted) {
    return false;
  }
  int numVals = 0;
  for (int j = 0; j < a.numAccs; ++j) {
    numVals += a.accs[j].kind == ACC_COUNT ? 0 : 1;
  }
  return numVals <= 2 && (a.capacity & (a.capacity - 1)) == 0 && a.capacity >= (1ULL << 16);
}

bool radixEligible(const vx355_agg& h, const AggArgs& a) {
  const bool hashed = radixHashedEligible(h, a);
  if (h.radixMinRows < 0 || (a.mode != MODE_ARRAY && !hashed) || a.rowList || a.rescanOld || a.numAccs < 1 ||
      a.numAccs > kRadixMaxAccs || a.numRows < h.radixMinRows || a.capacity > (1ULL << kRadixKeyBits) ||
      a.numRows > (1LL << radixRowBits(a.capacity))) {
    return false;
  }
  if (hashed) {
    return true;  // partitions follow the rows of the chunk (radixHashedShift), two levels reach 2^20 of them
  }
  const int shiftB = radixShiftB(radixWords(a));
  const uint64_t parts = (a.capacity + (1ULL << shiftB) - 1) >> shiftB;
  // Worth it when rows outnumber partitions by far and two levels reach every partition.
  return parts >= 2 && parts <= static_cast<uint64_t>(h.radixMaxBins) * kRadixMaxBins &&
      static_cast<uint64_t>(a.numRows) >= parts * 1024;
}

// High-cardinality array-mode chunk: partition the rows by group range (one or
// two LDS-histogram passes), then aggregate each partition in LDS.
void launchRadix(vx355_agg& h, AggArgs& a) {
  auto& rt = Runtime::get();
  const int64_t n = a.numRows;
  RadixArgs r{};
  r.a = a;
  const bool hashed = a.mode == MODE_NORMALIZED;
  const bool dense = hashed && h.denseNext;
  h.denseNext = false;
  // dense: no table to address - the "home slots" are those of a virtual table of
  // partitions x kHashSlots slots, so that a partition maps onto the LDS table one to one
  const uint64_t slotSpace = dense ? denseSlotSpace(n) : a.capacity;
  r.hashed = hashed ? 1 : 0;
  r.slotMask = slotSpace - 1;
  r.shiftB = hashed ? radixHashedShift(slotSpace, n) : radixShiftB(radixWords(a));
  uint64_t parts = (slotSpace + (1ULL << r.shiftB) - 1) >> r.shiftB;   // (level 2 may be coarsened below: k_rp_bucket_sample)
  // One level while the fan-out fits the LDS cursors (measured: 2400 bins in one
  // pass beat 64 x 64 in two); otherwise two balanced levels.
  const uint64_t maxBins1 = hashed ? kSortBins : static_cast<uint64_t>(h.radixMaxBins);
  r.shift2 = parts <= maxBins1 ? 0 : std::max(log2Ceil((parts + maxBins1 - 1) / maxBins1), log2Ceil(parts) / 2);
  r.numBins = static_cast<int32_t>((parts + (1ULL << r.shift2) - 1) >> r.shift2);
  r.keyBits = radixKeyBits(slotSpace);
  r.rowBits = radixRowBits(slotSpace);
  int32_t bins2 = 1 << r.shift2;
  for (int j = 0; j < a.numAccs; ++j) {
    r.valIdx[j] = a.accs[j].kind == ACC_COUNT ? -1 : r.numVals++;
    if (r.valIdx[j] >= 0) {
      r.accOfVal[r.valIdx[j]] = j;
    }
  }
  r.recWords = 1 + r.numVals + (hashed ? 1 : 0);   // hashed: the full key travels in the last word
  int64_t tileRows = h.radixTileRows > 0 ? h.radixTileRows : std::max<int64_t>(32768, ceilDiv(n, 2048));
  tileRows = (tileRows + 1023) & ~1023LL;
  r.tileRows = tileRows;
  r.numTiles = ceilDiv(n, tileRows);
  const int64_t cells1 = static_cast<int64_t>(r.numBins) * r.numTiles;
  uint32_t tileRecs = 65536;
  if (const char* e = std::getenv("VX355_AGG_RADIX_TILE2")) {
    tileRecs = static_cast<uint32_t>(std::max(1024, std::atoi(e)));
  }
  const int64_t maxTiles2 = ceilDiv(n, tileRecs) + r.numBins;
  const int64_t cells2 = r.shift2 ? static_cast<int64_t>(bins2) * maxTiles2 : 0;
  // (the record buffers are sized where the layout of each level is known: an optimistic level
  // takes regions of 1.25 - 1.5 x the records, and growing a 24 GB buffer to 30 GB on every
  // operator would cycle 54 GB through the block cache per level)
  const size_t recBytes = static_cast<size_t>(n) * r.recWords * 8 + 64;
  h.rpHist.ensure(static_cast<size_t>(std::max(cells1, cells2)) * 4 + 64);
  // offsets: level 1 [cells1 + 1], level 2 [cells2 + 1]
  h.rpOffsets.ensure(static_cast<size_t>(cells1 + 1 + cells2 + 1) * 8 + 64);
  uint64_t* offsets1 = h.rpOffsets.as<uint64_t>();
  uint64_t* offsets2 = offsets1 + cells1 + 1;
  r.hist = h.rpHist.as<uint32_t>();
  r.offsets = offsets1;
  const int grid1 = static_cast<int>(std::min<int64_t>(r.numTiles, rt.numCUs * 2));
  // Shape of pass 1: flat single integer key? flat 8-byte operands?
  int kw = 0;
  if (a.numKeys == 1 && a.numTerms == 0 && !a.ignoreNullKeys) {
    const ColView& kc = a.keys[0].col;
    if (kc.enc == VX355_FLAT && !kc.nulls && a.keys[0].range.multiplier == 1 &&
        (kc.kind == VX355_BIGINT || kc.kind == VX355_INTEGER)) {
      kw = kc.kind == VX355_BIGINT ? 8 : 4;
    }
  }
  bool flatV = kw != 0;
  for (int j = 0; j < a.numAccs; ++j) {
    const AccArg& acc = a.accs[j];
    if (acc.hasMask || acc.inProj >= 0) {
      flatV = false;
    } else if (acc.kind == ACC_COUNT) {
      flatV = flatV && !acc.hasIn;
    } else {
      const bool wantInt = acc.kind == ACC_SUM_I64 || acc.kind == ACC_SUM_I64_WRAP ||
          ((acc.kind == ACC_MIN || acc.kind == ACC_MAX) && acc.inIsInt);
      flatV = flatV && acc.hasIn && acc.in.enc == VX355_FLAT && !acc.in.nulls &&
          acc.in.kind == (wantInt ? VX355_BIGINT : VX355_DOUBLE);
    }
  }
  const bool sorted1 = h.radixSorted && r.numBins <= kSortBins;
  // Optimistic level 1 (two-level launches over a flat integer key or over hashed home slots): no
  // pass over the keys to count the bins - every bin that can be hit (the observed key range; all
  // of them for home slots) owns a region of 1.25 x the even share + 4096 records and a cursor. A bin
  // that outgrows its region (keys bunched inside the range) sends the level back to counting.
  bool opt1 = h.radixOptimistic1 && h.radixOptimistic && sorted1 && r.shift2 != 0 && (hashed || kw != 0) && flatV;
  int64_t firstLive = 0, liveBins = r.numBins;
  if (opt1 && !hashed) {
    const auto& ks = h.keys[0];
    const int shift = r.shiftB + r.shift2;
    const uint64_t idMin = static_cast<uint64_t>(ks.obsMin) - static_cast<uint64_t>(ks.range.min) + 1;
    const uint64_t idMax = static_cast<uint64_t>(ks.obsMax) - static_cast<uint64_t>(ks.range.min) + 1;
    firstLive = static_cast<int64_t>(idMin >> shift);
    liveBins = static_cast<int64_t>(idMax >> shift) - firstLive + 1;
    opt1 = ks.hasObserved && liveBins >= 1 && firstLive + liveBins <= r.numBins;
  }
  uint32_t* binCursor = nullptr;
  uint64_t* binFirst = nullptr;
  auto exactLevel1 = [&]() {
    h.rpRecs1.ensure(recBytes);
    r.recs = h.rpRecs1.as<uint64_t>();
    if (kw == 8) {
      VX_LAUNCH("k_rp_count1", k_rp_count1<8>, grid1, 1024, 0, r);
    } else if (kw == 4) {
      VX_LAUNCH("k_rp_count1", k_rp_count1<4>, grid1, 1024, 0, r);
    } else {
      VX_LAUNCH("k_rp_count1", k_rp_count1<0>, grid1, 1024, 0, r);
    }
    scanU32ToU64(r.hist, cells1, offsets1, h.rpScan);
  };
  if (opt1) {
    r.binCap = static_cast<uint64_t>(n / liveBins + n / liveBins / 4 + 4096);
    // (the cursors are 32-bit: a region never holds more than that)
    opt1 = r.binCap < (1ULL << 32);
  }
  // Compact records (recLoad): nobody wants first-seen order, one operand, word 0 fits 32 bits without the
  // row, and both levels run their optimistic form (an overflow anywhere restarts the chunk with 16-byte records).
  const bool cr = opt1 && h.compactRecords && h.unorderedOutput && !hashed && r.recWords == 2 && (kw == 8 || kw == 4) &&
      r.keyBits + kRadixMaskBits <= 32 && h.radixOptimistic && bins2 <= kSortBins;
  if (cr) {
    r.rowBits = 0;
    r.crCap = static_cast<uint64_t>(liveBins) * r.binCap + 64;
  }
  // Keyed records (hashRecLoad): the same wish over an open-addressing table - flat BIGINT key, one flat operand
  // (every record has the same accumulator mask): {key, operand}, the home slot recomputed where it is needed.
  const bool kr = opt1 && h.compactRecords && h.unorderedOutput && hashed && flatV && kw == 8 && r.numVals == 1 &&
      h.radixOptimistic && bins2 <= kSortBins;
  if (kr) {
    r.recWords = 2;
  }
  if (opt1) {
    h.rpRecs1.ensure((static_cast<size_t>(liveBins) * r.binCap + 64) * r.recWords * 8 + 64);
    r.recs = h.rpRecs1.as<uint64_t>();
    char* lay = static_cast<char*>(h.rpLayout1.ensure(static_cast<size_t>(r.numBins) * 12 + 64 + 64));
    binFirst = reinterpret_cast<uint64_t*>(lay);
    binCursor = reinterpret_cast<uint32_t*>(lay + static_cast<size_t>(r.numBins) * 8);
    std::vector<uint64_t> first(static_cast<size_t>(r.numBins), 0);
    for (int64_t b = 0; b < r.numBins; ++b) {
      const int64_t live = std::min<int64_t>(std::max<int64_t>(b - firstLive, 0), liveBins - 1);
      first[static_cast<size_t>(b)] = static_cast<uint64_t>(live) * r.binCap;  // (bins outside the range are never hit)
    }
    // cursors: 0 for the bins of the observed key range; a bin outside it (a key inside the PADDED
    // range, VectorHasher's 50 % reserve, that the statistics pass did not see) has no region: its
    // cursor starts beyond every capacity, so the first record sends the level back to counting
    std::vector<uint32_t> cursors(static_cast<size_t>(r.numBins) + 16, 0);
    for (int64_t b = 0; b < r.numBins; ++b) {
      if (b < firstLive || b >= firstLive + liveBins) {
        cursors[static_cast<size_t>(b)] = kDeadBinCursor;
      }
    }
    rt.sync();  // (the previous launch's staging of these small tables must be over)
    copyIn(binFirst, first.data(), VX355_MEM_HOST, first.size() * 8);
    copyIn(binCursor, cursors.data(), VX355_MEM_HOST, cursors.size() * 4);
    rt.sync();
    r.binFirst = binFirst;
    r.binCursor = binCursor;
    r.binOverflow = binCursor + r.numBins;
  } else {
    exactLevel1();
  }
  auto scatter1 = [&](auto wTag) {
    constexpr int W = decltype(wTag)::value;
    if (sorted1 && opt1) {
      const int grid = static_cast<int>(std::min<int64_t>(r.numTiles, rt.numCUs * 2));
      if constexpr (W == 2) {
        if (kr) {
          VX_LAUNCH("k_rp_scatter1", (k_rp_scatter1_sorted<8, 2, true, true, true, false, true>), grid, kSortThreads, 0, r);
          return;
        }
      }
      if constexpr (W >= 2) {
        if (hashed) {
          if (kw == 8) {
            VX_LAUNCH("k_rp_scatter1", (k_rp_scatter1_sorted<8, W, true, true, true>), grid, kSortThreads, 0, r);
          } else {
            VX_LAUNCH("k_rp_scatter1", (k_rp_scatter1_sorted<0, W, false, true, true>), grid, kSortThreads, 0, r);
          }
          return;
        }
      }
      if constexpr (W == 2) {
        if (cr) {
          if (kw == 8) {
            VX_LAUNCH("k_rp_scatter1", (k_rp_scatter1_sorted<8, 2, true, false, true, true>), grid, kSortThreads, 0, r);
          } else {
            VX_LAUNCH("k_rp_scatter1", (k_rp_scatter1_sorted<4, 2, true, false, true, true>), grid, kSortThreads, 0, r);
          }
          return;
        }
      }
      if (kw == 8) {
        VX_LAUNCH("k_rp_scatter1", (k_rp_scatter1_sorted<8, W, true, false, true>), grid, kSortThreads, 0, r);
      } else {
        VX_LAUNCH("k_rp_scatter1", (k_rp_scatter1_sorted<4, W, true, false, true>), grid, kSortThreads, 0, r);
      }
      return;
    }
    if (sorted1) {
      const int grid = static_cast<int>(std::min<int64_t>(r.numTiles, rt.numCUs * 2));
      if constexpr (W >= 2) {
        if (hashed) {
          if (flatV && kw == 8) {
            VX_LAUNCH("k_rp_scatter1", (k_rp_scatter1_sorted<8, W, true, true>), grid, kSortThreads, 0, r);
          } else {
            VX_LAUNCH("k_rp_scatter1", (k_rp_scatter1_sorted<0, W, false, true>), grid, kSortThreads, 0, r);
          }
          return;
        }
      }
      if (flatV && kw == 8) {
        VX_LAUNCH("k_rp_scatter1", (k_rp_scatter1_sorted<8, W, true, false>), grid, kSortThreads, 0, r);
      } else if (flatV && kw == 4) {
        VX_LAUNCH("k_rp_scatter1", (k_rp_scatter1_sorted<4, W, true, false>), grid, kSortThreads, 0, r);
      } else {
        VX_LAUNCH("k_rp_scatter1", (k_rp_scatter1_sorted<0, W, false, false>), grid, kSortThreads, 0, r);
      }
      return;
    }
    if (flatV && kw == 8) {
      VX_LAUNCH("k_rp_scatter1", (k_rp_scatter1<8, W, true>), grid1, 1024, 0, r);
    } else if (flatV && kw == 4) {
      VX_LAUNCH("k_rp_scatter1", (k_rp_scatter1<4, W, true>), grid1, 1024, 0, r);
    } else {
      VX_LAUNCH("k_rp_scatter1", (k_rp_scatter1<0, W, false>), grid1, 1024, 0, r);
    }
  };
  auto byWidth = [&](auto&& fn) {
    switch (r.recWords) {
      case 1:
        fn(std::integral_constant<int, 1>{});
        break;
      case 2:
        fn(std::integral_constant<int, 2>{});
        break;
      case 3:
        fn(std::integral_constant<int, 3>{});
        break;
      default:
        fn(std::integral_constant<int, 4>{});
        break;
    }
  };
  byWidth(scatter1);
  if (opt1) {
    uint32_t full = 0;
    copyOut(&full, VX355_MEM_HOST, r.binOverflow, 4);
    if (full != 0 && (cr || kr)) {
      h.compactRecords = false;  // (the exact passes below know complete records only: start the chunk over)
      h.denseNext = dense;
      launchRadix(h, a);
      return;
    }
    if (full != 0) {
      // keys bunched inside the observed range: count, then scatter to exact offsets; this operator
      // counts from now on
      ++h.radixRedone;
      h.radixOptimistic1 = false;
      opt1 = false;
      r.binFirst = nullptr;
      r.binCursor = nullptr;
      exactLevel1();
      byWidth(scatter1);
    }
  }
  // Coarser level 2 when the keys repeat (k_rp_bucket_sample): hashed, both levels optimistic.
  if (hashed && opt1 && h.coarseLevel2 && h.hashSlotsFixed == 0 && h.radixOptimistic && r.shift2 >= 3 &&
      r.numBins >= kBucketSamples && bins2 <= kSortBins) {
    unsigned long long* sets = static_cast<unsigned long long*>(
        h.rpSampleSets.ensure(static_cast<size_t>(kBucketSamples) * kBucketSet * 8 + kBucketSamples * 8 + 64));
    uint32_t* perBucket = reinterpret_cast<uint32_t*>(sets + static_cast<size_t>(kBucketSamples) * kBucketSet);
    HIP_OK(hipMemsetAsync(sets, 0xff, static_cast<size_t>(kBucketSamples) * kBucketSet * 8, rt.stream));
    HIP_OK(hipMemsetAsync(perBucket, 0, kBucketSamples * 8, rt.stream));
    const uint64_t keyMask = (1ULL << r.keyBits) - 1;
    const int gridS = kBucketSamples * kBucketSlices;
    if (kr) {
      VX_LAUNCH("k_rp_bucket_sample", (k_rp_bucket_sample<2, true>), gridS, 1024, 0, r.recs, binFirst, binCursor, r.numBins,
                r.shiftB, r.shift2, r.slotMask, keyMask, sets, perBucket);
    } else {
      byWidth([&](auto wTag) {
        constexpr int W = decltype(wTag)::value;
        if constexpr (W >= 2) {
          VX_LAUNCH("k_rp_bucket_sample", (k_rp_bucket_sample<W, false>), gridS, 1024, 0, r.recs, binFirst, binCursor,
                    r.numBins, r.shiftB, r.shift2, r.slotMask, keyMask, sets, perBucket);
        }
      });
    }
    uint32_t counts[2 * kBucketSamples];
    copyOut(counts, VX355_MEM_HOST, perBucket, sizeof(counts));
    uint64_t found = 0, records = 0;
    uint32_t fullest = 0;
    for (int b = 0; b < kBucketSamples; ++b) {
      found += counts[2 * b];
      records += counts[2 * b + 1];
      fullest = std::max(fullest, counts[2 * b]);
    }
    if (records >= 1024) {
      // keys of a partition of the CURRENT size: the fullest one seen, or the mean with room for its spread
      const double mean = static_cast<double>(found) / kBucketSamples;
      const double perPart = std::max<double>(fullest, mean + 6.0 * std::sqrt(mean + 1.0));
      int coarse = 0;
      // (a partition that is 2^coarse times larger holds that many times the keys: they must fit the LDS table at
      // load <= 0.4, and the chip must still see >= 16 partitions per CU)
      while (coarse < 3 && r.shift2 - (coarse + 1) >= 2 && 2.5 * perPart * (2 << coarse) <= kHashSlots &&
             (parts >> (coarse + 1)) >= static_cast<uint64_t>(rt.numCUs) * 16) {
        ++coarse;
      }
      if (coarse > 0) {
        r.shiftB += coarse;
        r.shift2 -= coarse;
        bins2 >>= coarse;
        parts >>= coarse;
        ++h.coarsenedLaunches;
      }
    }
  }
  const Level1Bins level1{offsets1, r.numTiles, opt1 ? binFirst : nullptr, opt1 ? binCursor : nullptr};

  RadixAggArgs g{};
  int32_t hashSlots = h.hashSlotsFixed > 0 ? h.hashSlotsFixed : kHashSlots;
  int32_t groupShift = 0;   // hashed folds: log2(partitions per LDS table and flush), from the key sample below
  g.recs = h.rpRecs1.as<uint64_t>();
  g.partBegin = offsets1;
  g.partCell = nullptr;
  g.cellStride = r.numTiles;
  if (r.shift2) {
    h.rpTiles.ensure(static_cast<size_t>(maxTiles2) * sizeof(RadixTile) + 64);
    h.rpMisc.ensure(64 + static_cast<size_t>(parts + 1) * 4 + 64);
    uint32_t* numTiles2 = h.rpMisc.as<uint32_t>();
    uint32_t* partCell = numTiles2 + 16;
    VX_LAUNCH("k_rp_tiles", k_rp_tiles, static_cast<int>(std::max<uint64_t>(1, std::min<uint64_t>(64, parts >> 14))), 1024, 0,
              level1, r.numBins, bins2, r.shift2, tileRecs,
              h.rpTiles.as<RadixTile>(), numTiles2, partCell, static_cast<int64_t>(parts));
    Radix2Args r2{};
    r2.in = h.rpRecs1.as<uint64_t>();
    r2.out = h.rpRecs2.as<uint64_t>();
    r2.tiles = h.rpTiles.as<RadixTile>();
    r2.numTiles = numTiles2;
    r2.recWords = r.recWords;
    r2.shiftB = r.shiftB;
    r2.numBins = bins2;
    r2.hist = h.rpHist.as<uint32_t>();
    r2.offsets = offsets2;
    const int grid2 = static_cast<int>(std::min<int64_t>(maxTiles2, rt.numCUs * 2));
    bool exact = true;
    if (h.radixOptimistic && h.radixSorted && bins2 <= kSortBins) {
      // no counting pass: regions of 1.5 x the even share per partition, claimed with atomics
      const size_t partsPadded = static_cast<size_t>(r.numBins) * bins2;
      char* lay = static_cast<char*>(h.rpLayout2.ensure((partsPadded + 1) * 8 + static_cast<size_t>(r.numBins) * 4 +
                                                        partsPadded * 4 + 64 + 64));
      uint64_t* partBase = reinterpret_cast<uint64_t*>(lay);
      uint32_t* bucketCap = reinterpret_cast<uint32_t*>(lay + (partsPadded + 1) * 8);
      uint32_t* partCount = bucketCap + r.numBins;
      uint32_t* overflow = partCount + partsPadded;   // [0] flag, [2..3] total records of the layout
      HIP_OK(hipMemsetAsync(partCount, 0, (partsPadded + 16) * 4, rt.stream));
      VX_LAUNCH("k_rp_layout2", k_rp_layout2, 1, 1024, 0, level1, r.numBins, r.shift2,
                static_cast<int64_t>(partsPadded), partBase, bucketCap, reinterpret_cast<uint64_t*>(overflow + 2));
      uint64_t layoutRecs = 0;
      copyOut(&layoutRecs, VX355_MEM_HOST, overflow + 2, 8);
      h.rpRecs2.ensure((static_cast<size_t>(layoutRecs) + 64) * r.recWords * 8 + 64);
      r2.out = h.rpRecs2.as<uint64_t>();
      Radix2OptArgs o{};
      o.in = r2.in;
      o.out = r2.out;
      o.tiles = r2.tiles;
      o.numTiles = numTiles2;
      o.shiftB = r.shiftB;
      o.shift2 = r.shift2;
      o.numBins = bins2;
      o.keyBits = r.keyBits;
      o.partBase = partBase;
      o.bucketCap = bucketCap;
      o.partCount = partCount;
      o.overflow = overflow;
      o.crCapIn = r.crCap;
      o.crCapOut = static_cast<uint64_t>(layoutRecs) + 64;
      o.slotMask = r.slotMask;
      if (cr) {
        VX_LAUNCH("k_rp_scatter2", (k_rp_scatter2_opt<2, true>), grid2, kSortThreads, 0, o);
      } else if (kr) {
        VX_LAUNCH("k_rp_scatter2", (k_rp_scatter2_opt<2, false, true>), grid2, kSortThreads, 0, o);
      } else {
        byWidth([&](auto wTag) {
          VX_LAUNCH("k_rp_scatter2", (k_rp_scatter2_opt<decltype(wTag)::value>), grid2, kSortThreads, 0, o);
        });
      }
      constexpr int kSampleParts = 64;
      const bool sampleKeys = hashed && h.hashSlotsFixed == 0 && partsPadded >= kSampleParts;
      if (sampleKeys) {
        byWidth([&](auto wTag) {
          constexpr int W = decltype(wTag)::value;
          VX_LAUNCH("k_rp_distinct_sample", (k_rp_distinct_sample<W>), kSampleParts, 256, 0, o.out, partBase, partCount,
                    static_cast<int64_t>(partsPadded), overflow + 4, kr ? 0 : W - 1);
        });
      }
      uint32_t flags[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // [0] overflow, [4..6] the key sample
      copyOut(flags, VX355_MEM_HOST, overflow, sizeof(flags));
      exact = flags[0] != 0;  // some partition outgrew its region (skewed keys): redo the level exactly
      if (exact && (cr || kr)) {
        h.compactRecords = false;  // (as at level 1: the exact passes read complete records)
        h.denseNext = dense;
        launchRadix(h, a);
        return;
      }
      if (!exact) {
        g.crCap = cr ? o.crCapOut : 0;
        g.krSlotMask = kr ? r.slotMask : 0;
        g.krMask = static_cast<uint32_t>((1u << a.numAccs) - 1);
        g.partBase = partBase;
        g.partCount = partCount;
        if (sampleKeys && flags[5] != 0) {
          // LDS entries of a fold: load <= 0.4 for the fullest sampled partition's keys scaled to a
          // whole partition (the sample looks at 2048 records at most), with room for their spread
          const double perRecord = static_cast<double>(flags[4]) / static_cast<double>(flags[5]);
          const double recsPerPart = static_cast<double>(n) / static_cast<double>(parts);
          const double mean = perRecord * recsPerPart;
          const double fullest = std::max<double>(flags[6], mean + 6.0 * std::sqrt(mean + 1.0));
          hashSlots = static_cast<int32_t>(nextPow2(static_cast<uint64_t>(std::min<double>(kHashSlots, std::max(512.0, 2.5 * fullest)))));
          // ... and few keys per partition share a table: as many partitions per flush as the largest table takes
          // (config 4 with sparse keys, ~95 keys per partition, fold alone: 11.0 ms one partition per flush, 10.0
          // with two in 1024 entries, 9.4 with four in 2048 - fewer workgroups per CU, still faster)
          int32_t groupMax = kHashSlots;
          if (const char* e = std::getenv("VX355_AGG_FOLD_GROUP_SLOTS")) {
            groupMax = std::max(512, std::min(kHashSlots, std::atoi(e)));
          }
          while (h.foldGroups && groupShift < 4 && 2.5 * fullest * (2 << groupShift) <= groupMax &&
                 (parts >> (groupShift + 1)) >= static_cast<uint64_t>(rt.numCUs) * 16) {
            ++groupShift;
          }
          if (groupShift > 0) {
            hashSlots = static_cast<int32_t>(nextPow2(static_cast<uint64_t>(
                std::min<double>(kHashSlots, std::max(512.0, 2.5 * fullest * (1 << groupShift))))));
          }
        }
      } else {
        ++h.radixRedone;
        h.radixOptimistic = false;  // skewed keys: this operator counts from now on
      }
    }
    if (exact) {
      h.rpRecs2.ensure(recBytes);
      r2.out = h.rpRecs2.as<uint64_t>();
      HIP_OK(hipMemsetAsync(r2.hist, 0, static_cast<size_t>(cells2) * 4, rt.stream));
      VX_LAUNCH("k_rp_count2", k_rp_count2, grid2, 1024, 0, r2);
      scanU32ToU64(r2.hist, cells2, offsets2, h.rpScan);
      byWidth([&](auto wTag) {
        if (h.radixSorted && bins2 <= kSortBins) {
          VX_LAUNCH("k_rp_scatter2", (k_rp_scatter2_sorted<decltype(wTag)::value>), grid2, kSortThreads, 0, r2);
        } else {
          VX_LAUNCH("k_rp_scatter2", (k_rp_scatter2<decltype(wTag)::value>), grid2, 1024, 0, r2);
        }
      });
      g.partBegin = offsets2;
      g.partCell = partCell;
    }
    g.recs = h.rpRecs2.as<uint64_t>();
  }
  g.numParts = static_cast<int64_t>(parts);
  g.splitList = static_cast<uint32_t*>(h.rpSplit.ensure((16 + static_cast<size_t>(parts)) * 4 + 64));
  HIP_OK(hipMemsetAsync(g.splitList, 0, 64, rt.stream));
  g.recWords = r.recWords;
  g.numAccs = a.numAccs;
  g.shiftB = r.shiftB;
  g.stride = a.stride;
  g.table = a.table;
  g.rowBase = a.rowBase;
  g.capacity = a.capacity;
  g.counters = a.counters;
  for (int j = 0; j < a.numAccs; ++j) {
    g.kind[j] = a.accs[j].kind;
    g.off[j] = a.accs[j].off;
    g.valIdx[j] = r.valIdx[j];
    g.accOfVal[j] = r.accOfVal[j];
    g.splitM[j] = a.accs[j].splitM;
    g.ldsIdx[j] = g.numWords;
    g.wordKind[g.numWords] = a.accs[j].kind;
    g.wordOff[g.numWords] = a.accs[j].off;
    ++g.numWords;
    if (accWords(a.accs[j].kind) == 2) {
      g.wordKind[g.numWords] = accSecondKind(a.accs[j].kind);
      g.wordOff[g.numWords] = a.accs[j].off + 1;
      ++g.numWords;
    }
  }
  // LDS of one fold: dense = B groups x (words + first row); hashed = (B + margin) window entries x
  // (key + words + first row)
  g.hashSlots = hashSlots;
  g.groupShift = groupShift;
  h.lastHashSlots = hashed ? hashSlots : 0;
  const size_t ldsBytes = hashed ? static_cast<size_t>(hashSlots) * (8 + g.numWords * 8 + 4)
                                 : (static_cast<size_t>(1) << r.shiftB) * (g.numWords * 8 + 4);
  // Few partitions or skewed keys: slices keep every CU busy.
  g.sliceRecs = static_cast<uint64_t>(std::max<int64_t>(1 << 16, ceilDiv(n, 2048)));
  if (const char* e = std::getenv("VX355_AGG_RADIX_SLICE")) {
    g.sliceRecs = static_cast<uint64_t>(std::max<int64_t>(512, std::strtoll(e, nullptr, 10)));
  }
  g.keyBits = r.keyBits;
  g.rowBits = r.rowBits;
  g.virgin = h.tableVirgin ? 1 : 0;
  g.pattern = h.pattern.as<uint64_t>();
  for (int w = 0; w < 2 + kMaxLdsAccs; ++w) {
    g.ldsOfWord[w] = -1;
  }
  for (int j = 0; j < g.numWords; ++j) {
    g.ldsOfWord[g.wordOff[j]] = static_cast<int8_t>(j);
  }
  // dense: the folds append to a fresh array of group rows, sized for "a quarter of the rows are
  // new groups" first and exactly on the second try (the folds count every group they find)
  DevBuf denseRows;
  uint64_t denseCap = 0;
  if (dense) {
    denseCap = static_cast<uint64_t>(std::min<int64_t>(n, std::max<int64_t>(1 << 20, n / 4)));
    if (const char* e = std::getenv("VX355_AGG_DENSE_CAP")) {
      denseCap = static_cast<uint64_t>(std::max<int64_t>(64, std::strtoll(e, nullptr, 10)));
    }
    denseRows.ensure(static_cast<size_t>(denseCap) * a.stride * 8 + 64);
    h.denseFlags.ensure(64);
    HIP_OK(hipMemsetAsync(h.denseFlags.ptr(), 0, 64, rt.stream));
    g.dense = 1;
    g.denseCap = denseCap;
    g.denseFlags = h.denseFlags.as<uint32_t>();
    // rows a workgroup takes at a time: the holes stay ~1 % of the array
    g.denseChunk = static_cast<uint32_t>(std::max<uint64_t>(256, std::min<uint64_t>(8192, denseCap / (static_cast<uint64_t>(rt.numCUs) * 16))));
    if (const char* e = std::getenv("VX355_AGG_DENSE_CHUNK")) {
      g.denseChunk = static_cast<uint32_t>(std::max(1, std::atoi(e)));
    }

    g.table = denseRows.as<uint64_t>();
    h.pairsComplete = true;  // no groups yet: this launch lists all of them
    h.pairCount = 0;
    h.pairHoles = 0;
  }
  if (h.pairsComplete && h.pairCount == h.numGroups) {
    // room for one new group per row of the chunk, at most one per group row
    const size_t room = dense ? static_cast<size_t>(denseCap)
                              : static_cast<size_t>(std::min<uint64_t>(a.capacity, static_cast<uint64_t>(h.numGroups + n)));
    const size_t live = static_cast<size_t>(h.pairCount + h.pairHoles);
    h.orderKeys.ensure((room + static_cast<size_t>(h.pairHoles)) * 8 + 64, true, live * 8);
    h.orderVals.ensure((room + static_cast<size_t>(h.pairHoles)) * 4 + 64, true, live * 4);
    g.pairKeys = h.orderKeys.as<uint64_t>();
    g.pairVals = h.orderVals.as<uint32_t>();
    g.pairBase = static_cast<uint64_t>(live);
  } else {
    h.pairsComplete = false;
  }
  // workgroups per CU: as many as the LDS holds, at most 4 (a fold alternates between streaming
  // records in and flushing its partition: the other workgroups of the CU cover those phases)
  int perCu = static_cast<int>(std::min<size_t>(4, (150 * 1024) / (ldsBytes + 2304)));
  if (const char* e = std::getenv("VX355_AGG_FOLD_WGS")) {
    perCu = std::atoi(e);
  }
  const int gridA = rt.numCUs * std::max(1, perCu);
  if (hashed) {
    auto fold = [&]() {
      HIP_OK(hipMemsetAsync(g.splitList, 0, 64, rt.stream));
      for (int phase = 0; phase < 2; ++phase) {   // the owners, then the other slices of split partitions
        g.phase = phase;
        byWidth([&](auto wTag) {
          constexpr int W = decltype(wTag)::value;
          if constexpr (W == 2) {
            if (g.krSlotMask != 0) {
              if (dense) {
                VX_LAUNCH("k_rp_aggregate", (k_rp_aggregate_hashed<3, true, true>), gridA, 512, ldsBytes, g);
              } else {
                VX_LAUNCH("k_rp_aggregate", (k_rp_aggregate_hashed<3, false, true>), gridA, 512, ldsBytes, g);
              }
              return;
            }
          }
          if constexpr (W >= 2) {
            if (dense) {
              VX_LAUNCH("k_rp_aggregate", (k_rp_aggregate_hashed<W, true>), gridA, 512, ldsBytes, g);
            } else {
              VX_LAUNCH("k_rp_aggregate", (k_rp_aggregate_hashed<W, false>), gridA, 512, ldsBytes, g);
            }
          }
        });
      }
    };
    fold();
    ++h.radixLaunches;
    h.compactLaunches += g.krSlotMask != 0 ? 1 : 0;
    if (!dense) {
      return;
    }
    ++h.denseLaunches;
    uint32_t* newGroups = &a.counters->numNewGroups;
    uint32_t flags[2] = {0, 0};   // [0] some key may own several rows, [1] rows handed out (groups + holes)
    copyOut(flags, VX355_MEM_HOST, h.denseFlags.ptr(), 8);
    while (flags[1] > denseCap) {
      // more rows than the guess: now the number is known (the holes vary a little from run to run:
      // hence the margin), fold the same records again
      ++h.denseRefolds;
      denseCap = static_cast<uint64_t>(flags[1]) + flags[1] / 16 + 65536;
      denseRows.ensure(static_cast<size_t>(denseCap) * a.stride * 8 + 64);
      h.orderKeys.ensure(static_cast<size_t>(denseCap) * 8 + 64);
      h.orderVals.ensure(static_cast<size_t>(denseCap) * 4 + 64);
      g.pairKeys = h.orderKeys.as<uint64_t>();
      g.pairVals = h.orderVals.as<uint32_t>();
      g.denseCap = denseCap;
      g.table = denseRows.as<uint64_t>();
      HIP_OK(hipMemsetAsync(newGroups, 0, 4, rt.stream));
      HIP_OK(hipMemsetAsync(h.denseFlags.ptr(), 0, 64, rt.stream));
      fold();
      copyOut(flags, VX355_MEM_HOST, h.denseFlags.ptr(), 8);
    }
    const uint32_t found = flags[1];
    uint32_t groups = 0;
    copyOut(&groups, VX355_MEM_HOST, newGroups, 4);
    h.pairHoles = static_cast<int64_t>(found) - static_cast<int64_t>(groups);  // the caller adds the groups to pairCount
    if (flags[0] == 0) {
      // the array of rows is the table from here on
      h.table = std::move(denseRows);
      h.capacity = found;
      h.tableDense = true;
      h.tableVirgin = false;
      h.tableReady = true;
      return;
    }
    // some key may own several rows (a partition folded in slices): merge them in a real table
    ++h.denseMerges;
    const uint64_t cap = hashCapacityFor(static_cast<uint64_t>(found));
    DevBuf fresh;
    initTable(h, fresh, cap);
    HIP_OK(hipMemsetAsync(newGroups, 0, 4, rt.stream));
    DenseMergeArgs m{};
    m.rows = denseRows.as<uint64_t>();
    m.numRows = found;
    m.table = fresh.as<uint64_t>();
    m.capacity = cap;
    m.stride = a.stride;
    m.numWords = g.numWords;
    for (int j = 0; j < g.numWords; ++j) {
      m.wordKind[j] = g.wordKind[j];
      m.wordOff[j] = g.wordOff[j];
    }
    m.counters = a.counters;
    VX_LAUNCH("k_dense_merge", k_dense_merge, streamGrid(static_cast<int64_t>(found), 256), 256, 0, m);
    rt.sync();
    h.table = std::move(fresh);
    h.capacity = cap;
    h.tableDense = false;
    h.tableVirgin = false;
    h.tableReady = true;
    h.pairsComplete = false;
    h.pairHoles = 0;
    return;
  }
  // Two launches: the owners of the partitions (a virgin table: they store complete rows), then the
  // other slices of the partitions the owners listed as split (nothing to do for evenly spread keys).
  auto foldLaunch = [&]() {
    if (g.crCap != 0) {
      VX_LAUNCH("k_rp_aggregate", (k_rp_aggregate<2, true>), gridA, 512, ldsBytes, g);
    } else {
      byWidth([&](auto wTag) {
        VX_LAUNCH("k_rp_aggregate", (k_rp_aggregate<decltype(wTag)::value>), gridA, 512, ldsBytes, g);
      });
    }
  };
  g.phase = 0;
  foldLaunch();
  g.phase = 1;
  g.virgin = 0;
  foldLaunch();
  h.tableVirgin = false;
  ++h.radixLaunches;
  h.compactLaunches += g.crCap != 0 ? 1 : 0;
}

// Grid of an LDS launch and how it flushes. Atomics: enough rows per workgroup that the flush (live
// slots x words HBM atomics per workgroup, ~20 G/s chip-wide) stays a small fraction of the work.
// Scratch (direct-index tables whose flush would retire > 256 K atomics - BASELINE config 1's 1000
// groups x 4 words x 800 workgroups = 3 M, 0.15 ms on a 27-us problem): as many workgroups as fill
// the chip, every one stores capacity x (words + 1) values, k_lds_reduce folds them.
int ldsGrid(vx355_agg& h, LdsPlan& plan, size_t ldsBytes, int64_t rows, int threads, int unroll, int maxBlocksPerCu) {
  auto& rt = Runtime::get();
  int blocksPerCu = std::max<int>(1, std::min<int>(maxBlocksPerCu, static_cast<int>((150 * 1024) / ldsBytes)));
  if (h.ldsBlocksPerCuCap > 0) {
    blocksPerCu = std::min(blocksPerCu, h.ldsBlocksPerCuCap);
  }
  const int64_t full = static_cast<int64_t>(rt.numCUs) * blocksPerCu;
  const int64_t fullScratch = static_cast<int64_t>(rt.numCUs) * std::min(blocksPerCu, h.scratchBlocksPerCu);
  const int64_t tile = static_cast<int64_t>(threads) * unroll;
  const int64_t minRowsPerBlock = std::max<int64_t>(tile, 4LL * plan.S * plan.A);
  const int64_t gridAtomic = std::max<int64_t>(1, std::min<int64_t>(ceilDiv(rows, minRowsPerBlock), full));
  plan.scratch = nullptr;
  if (h.ldsScratchFlush && plan.tableMode == MODE_ARRAY && plan.direct != 2) {
    // (compact slots before the first launch has counted the groups: what the first rows showed)
    const int64_t live = plan.direct == 1
        ? static_cast<int64_t>(plan.capacity)
        : std::min<int64_t>(plan.S, std::max<int64_t>({h.numGroups, h.firstRowsDistinct, 1}));
    const int64_t gridScratch = std::max<int64_t>(1, std::min<int64_t>(ceilDiv(rows, tile), fullScratch));
    const int64_t perCopy = static_cast<int64_t>(plan.capacity) * (plan.A + 1) * 8;
    if (live * (plan.A + 1) * gridAtomic > h.scratchMinAtomics && perCopy * gridScratch <= (64LL << 20)) {
      h.ldsScratch.ensure(static_cast<size_t>(perCopy * gridScratch) + 64);
      plan.scratch = h.ldsScratch.as<uint64_t>();
      return static_cast<int>(gridScratch);
    }
  }
  return static_cast<int>(gridAtomic);
}

// Before an LDS launch: the table must be initialised - unless the launch flushes through scratch
// copies into a table nobody has written (every word of every row is then stored by k_lds_reduce).
bool ldsPrepareTable(vx355_agg& h, const LdsPlan& plan) {
  // (direct slots only: with a compact map a workgroup that runs out of slots updates the table itself)
  const bool storeAll = plan.scratch != nullptr && h.tableVirgin && plan.A == h.stride - 2 && plan.direct == 1;
  if (!storeAll) {
    settleTable(h);
  }
  return storeAll;
}

void ldsReduce(vx355_agg& h, const LdsPlan& plan, int grid, bool storeAll) {
  if (plan.scratch == nullptr) {
    return;
  }
  const int64_t elements = static_cast<int64_t>(plan.capacity) * (plan.A + 1);
  const int tiles = static_cast<int>(ceilDiv(elements, 64));
  // few (word, key) tiles: several block columns share the copies so that the chip is busy
  int columns = storeAll ? 1 : std::max(1, std::min({4, grid / 64, (Runtime::get().numCUs * 2) / std::max(1, tiles)}));
  ++h.scratchFlushes;
  // (the last launch of the chunk: its last workgroup hands the counters to the host)
  h.countersPublished = tiles * columns <= kMaxPublishingBlocks;
  VX_LAUNCH("k_lds_reduce", k_lds_reduce, dim3(tiles, columns), 1024, 0, plan, grid, storeAll ? 1 : 0,
            h.countersPublished ? counterMail(h) : CounterMail{});
  h.tableVirgin = false;
}

void launchChunk(vx355_agg& h, AggArgs& a) {
  auto& rt = Runtime::get();
  size_t ldsBytes = 0;
  LdsArgs la{};
  // LDS / table words of the launch: a DOUBLE sum owns two (hi, lo).
  int numWords = 0;
  for (int j = 0; j < a.numAccs; ++j) {
    a.accs[j].ldsIdx = numWords;
    la.plan.kind[numWords] = a.accs[j].kind;
    la.plan.off[numWords] = a.accs[j].off;
    ++numWords;
    if (accWords(a.accs[j].kind) == 2) {
      la.plan.kind[numWords] = accSecondKind(a.accs[j].kind);
      la.plan.off[numWords] = a.accs[j].off + 1;
      ++numWords;
    }
  }
  if (chooseLds(h, numWords, &la.plan, &ldsBytes)) {
    h.pairsComplete = false;
    LdsPlan& plan = la.plan;
    plan.table = a.table;
    plan.stride = a.stride;
    plan.rowBase = a.rowBase;
    plan.counters = a.counters;
    plan.tableMode = a.mode;
    plan.deferOverflow = (h.slotsOnlyLaunch && a.mode == MODE_NORMALIZED && plan.direct == 2 && !a.rowList &&
                          !a.rescanOld) ? 1 : 0;
    FastArgs fa;
    FastSignature sig;
    const bool fastClass = !h.disableFast && buildFastArgs(a, plan, &fa, &sig);
    if (!fastClass && h.logShapes && !h.disableFast) {
      fprintf(stderr, "vx355: plan outside the specialised class (keys %d, terms %d, accumulators %d)\n", a.numKeys,
              a.numTerms, a.numAccs);
    }
    if (fastClass) {
      if (const FastEntry* e = findFastEntry(sig, h.fastUnroll)) {
        const int grid = ldsGrid(h, fa.plan, ldsBytes, a.numRows, 512, e->unroll, 4);
        const bool storeAll = ldsPrepareTable(h, fa.plan);
        e->launch(fa, grid, ldsBytes);
        ldsReduce(h, fa.plan, grid, storeAll);
        return;
      }
      if (h.logShapes && !h.shapeLogged) {
        // (the template arguments of FastShape after UNROLL, ready for kFastTable's VX_FAST_ENTRY_K)
        h.shapeLogged = true;
        fprintf(stderr,
                "vx355: plan shape outside the ahead-of-time table: FastShape<U, %d, %d, %d, %d, %d, %d, 0x%llxull, "
                "0x%llxull, 0x%xu, 0x%xu, 0x%llxull, 0x%xu, 0x%xu, 0x%llxull, 0x%xu>\n",
                sig.k0, sig.k1, sig.t0, sig.t1, sig.numLoads, sig.numAccs,
                static_cast<unsigned long long>(sig.accLo), static_cast<unsigned long long>(sig.accHi), sig.ind, sig.nul,
                static_cast<unsigned long long>(sig.accEx), sig.lk, sig.msk, static_cast<unsigned long long>(sig.ops),
                sig.kx);
      }
      if (hipFunction_t fn = h.jitEnabled ? jitFastKernel(sig, h.fastUnroll, h.logShapes, h.jitAsync) : nullptr) {
        const int grid = ldsGrid(h, fa.plan, ldsBytes, a.numRows, 512, h.fastUnroll, 4);
        const bool storeAll = ldsPrepareTable(h, fa.plan);
        launchJitFast(fn, fa, grid, ldsBytes);
        ldsReduce(h, fa.plan, grid, storeAll);
        ++h.jitLaunches;
        return;
      }
    }
    la.a = a;
    int grid = static_cast<int>(std::min<int64_t>(ceilDiv(a.numRows, 1024), rt.numCUs * 2));
    if (h.ldsScratchFlush) {
      const int scratchGrid = ldsGrid(h, plan, ldsBytes, a.numRows, 1024, 1, 2);
      grid = plan.scratch ? scratchGrid : grid;
    }
    const bool storeAll = ldsPrepareTable(h, plan);
    VX_LAUNCH("k_agg_lds", k_agg_lds, grid, 1024, ldsBytes, la);
    ldsReduce(h, plan, grid, storeAll);
  } else {
    if (h.denseNext || radixEligible(h, a)) {
      launchRadix(h, a);
      return;
    }
    settleTable(h);
    h.pairsComplete = false;
    // DOUBLE sums keep their hi/lo split here too (two HBM atomics per sum and
    // row): the <= 1 ULP bound must not depend on which kernel a chunk takes.
    VX_LAUNCH("k_agg_global", k_agg_global, streamGrid(a.numRows, 256), 256, 0, a);
  }
}

void checkCounters(const Counters& c) {
  if (c.overflow) {
    // SumAggregate.cpp:24 + vector/AggregationHook.h:126-135: checkedPlus.
    VX_THROW(VX355_EUSER, "integer overflow");
  }
  if (c.unmappable) {
    VX_THROW(VX355_EINTERNAL, "unmappable grouping key outside the mode switch");
  }
  if (c.tableFull) {
    VX_THROW(VX355_EINTERNAL, "aggregation table full");
  }
}

int keyStoreWords(int32_t kind) { return (isString(kind) || kind == VX355_TIMESTAMP) ? 2 : 1; }

// Grows the generic-mode arrays so that 'needGroups' dense ids fit; slots are
// kept at <= 50 % load (HashTable.h:946-956 asks for <= 70 %).
void ensureGenericCapacity(vx355_agg& h, uint64_t needGroups) {
  auto& rt = Runtime::get();
  const size_t live = static_cast<size_t>(h.numGroups);
  if (h.gKeyStore.empty()) {
    h.gKeyStore.resize(h.keys.size());
    h.gCounter.ensure(64);
    HIP_OK(hipMemsetAsync(h.gCounter.ptr(), 0, 64, rt.stream));
  }
  if (needGroups > h.gMaxGroups) {
    const uint64_t newMax = std::max<uint64_t>({needGroups, h.gMaxGroups + h.gMaxGroups / 2, 1024});
    for (size_t k = 0; k < h.keys.size(); ++k) {
      const size_t w = static_cast<size_t>(keyStoreWords(h.keys[k].kind)) * 8;
      h.gKeyStore[k].ensure(newMax * w + 64, true, live * w);
    }
    h.gNullStore.ensure(newMax * 8 + 64, true, live * 8);
    h.gHashStore.ensure(newMax * 8 + 64, true, live * 8);
    DevBuf fresh;
    initTable(h, fresh, newMax);
    if (h.tableReady && live > 0) {
      copyIn(fresh.ptr(), h.table.ptr(), VX355_MEM_DEVICE, live * h.stride * 8);
      ++h.numRehashes;
    }
    rt.sync();
    h.table = std::move(fresh);
    h.gMaxGroups = newMax;
    h.capacity = newMax;
    h.tableReady = true;
  }
  const uint64_t wantSlots = nextPow2(std::max<uint64_t>(2048, 2 * h.gMaxGroups));
  if (wantSlots > h.gSlotCap) {
    DevBuf fresh;
    fresh.ensure(wantSlots * 8 + 64);
    HIP_OK(hipMemsetAsync(fresh.ptr(), 0, wantSlots * 8, rt.stream));
    if (live > 0) {
      VX_LAUNCH("k_generic_rehash", k_generic_rehash, streamGrid(static_cast<int64_t>(live), 256), 256, 0,
                fresh.as<uint64_t>(), wantSlots - 1, h.gHashStore.as<uint64_t>(),
                static_cast<uint32_t>(live));
    }
    rt.sync();
    h.gSlots = std::move(fresh);
    h.gSlotCap = wantSlots;
  }
}

void shiftView(ColView& v, int64_t begin) {
  if (begin == 0 || v.enc == VX355_CONSTANT) {
    return;
  }
  // Chunks start on a multiple of 64 rows, so bitmaps shift by whole words.
  if (v.nulls) {
    v.nulls += begin >> 6;
  }
  if (v.enc == VX355_DICTIONARY) {
    v.indices += begin;
    return;
  }
  const int w = kindWidth(v.kind);
  if (w == 0) {
    v.values = static_cast<const uint64_t*>(v.values) + (begin >> 6);
  } else {
    v.values = static_cast<const char*>(v.values) + begin * w;
  }
}

void shiftArgs(AggArgs& c, int64_t begin) {
  for (int k = 0; k < c.numKeys; ++k) {
    shiftView(c.keys[k].col, begin);
  }
  for (int j = 0; j < c.numAccs; ++j) {
    if (c.accs[j].hasIn) {
      shiftView(c.accs[j].in, begin);
    }
    if (c.accs[j].hasMask) {
      shiftView(c.accs[j].mask, begin);
    }
  }
  for (int t = 0; t < c.numTerms; ++t) {
    shiftView(c.terms[t].col, begin);
  }
  for (int j = 0; j < c.numProj; ++j) {
    for (int f = 0; f < c.proj[j].numFactors; ++f) {
      if (c.proj[j].factors[f].hasCol) {
        shiftView(c.proj[j].factors[f].col, begin);
      }
    }
  }
}

// One pass of the generic kernel over 'count' rows of 'base' (whose column views
// already start at the first row), or over rowList[0, count) when given.
void runGeneric(vx355_agg& h, const AggArgs& base, int64_t count, uint64_t rowBase, const int32_t* rowList,
                bool skipInside) {
  ensureGenericCapacity(h, static_cast<uint64_t>(h.numGroups + count));
  GenericArgs ga{};
  ga.a = base;
  ga.skipInside = skipInside ? 1 : 0;
  AggArgs& c = ga.a;
  c.numRows = count;
  c.rowList = rowList;
  c.rescanOld = nullptr;
  c.rowBase = rowBase;
  c.table = h.table.as<uint64_t>();
  c.capacity = h.capacity;
  c.mode = MODE_ARRAY;  // group row = table + group id * stride
  GenericPart& g = ga.g;
  g.slots = h.gSlots.as<uint64_t>();
  g.slotMask = h.gSlotCap - 1;
  for (size_t k = 0; k < h.keys.size(); ++k) {
    g.keyStore[k] = h.gKeyStore[k].as<uint64_t>();
    g.keyWords[k] = keyStoreWords(h.keys[k].kind);
  }
  g.nullStore = h.gNullStore.as<uint64_t>();
  g.hashStore = h.gHashStore.as<uint64_t>();
  g.gidCounter = h.gCounter.as<uint32_t>();
  g.maxGroups = static_cast<uint32_t>(std::min<uint64_t>(h.gMaxGroups, 0xfffffffeULL));
  if (h.hasStringKeys) {
    // Worst case every long string of the chunk founds a group: make sure the newest arena
    // block can take them all (one small reduction over the key columns).
    auto& rt = Runtime::get();
    unsigned long long* cursor = static_cast<unsigned long long*>(h.strCursor.ensure(64));
    LongBytesArgs lb{};
    lb.numKeys = c.numKeys;
    for (int k = 0; k < c.numKeys; ++k) {
      lb.keys[k] = c.keys[k].col;
    }
    lb.numRows = count;
    lb.rowList = rowList;
    lb.total = cursor + 1;
    HIP_OK(hipMemsetAsync(cursor + 1, 0, 8, rt.stream));
    VX_LAUNCH("k_long_key_bytes", k_long_key_bytes, streamGrid(count, 256), 256, 0, lb);
    unsigned long long need = 0;
    copyOut(&need, VX355_MEM_HOST, cursor + 1, 8);
    if (h.strBlocks.empty() || h.strUsedHost + need > h.strCap) {
      h.strBlocks.emplace_back();
      h.strCap = std::max<uint64_t>(need, 64ULL << 20);
      h.strBlocks.back().ensure(static_cast<size_t>(h.strCap) + 64);
      h.strUsedHost = 0;
      HIP_OK(hipMemsetAsync(cursor, 0, 8, rt.stream));
    }
    h.strUsedHost += need;
    g.arenaBase = h.strBlocks.back().as<char>();
    g.arenaCursor = cursor;
    g.arenaCap = h.strCap;
  }
  resetCounters(h);
  VX_LAUNCH("k_agg_generic", k_agg_generic, streamGrid(count, 256), 256, 0, ga);
  Counters ctr = readCounters(h);
  if (ctr.unmappable) {
    VX_THROW(VX355_EINTERNAL, "unmappable grouping key in generic mode");
  }
  checkCounters(ctr);
  uint32_t ids = 0;
  copyOut(&ids, VX355_MEM_HOST, h.gCounter.ptr(), 4);
  h.numGroups = ids;
}

// How many distinct keys does the batch hold? 16384 rows spread evenly over it (sorted inputs must
// not fool the answer) into an LDS hash set; the count decides whether the LDS kernels take a
// wide-range table (chooseLds) or the radix / atomics paths do.
void sampleCardinality(vx355_agg& h, const AggArgs& a, int64_t n) {
  h.cardSampled = true;
  AggArgs c = a;
  c.numRows = std::min<int64_t>(n, 1 << 14);
  c.rowList = nullptr;
  c.rescanOld = nullptr;
  c.table = h.table.as<uint64_t>();
  c.capacity = h.capacity;
  c.mode = h.mode;
  for (int k = 0; k < c.numKeys; ++k) {
    c.keys[k].range = h.keys[k].range;
  }
  const int64_t step = std::max<int64_t>(1, n / c.numRows);
  resetCounters(h);
  uint32_t* out = reinterpret_cast<uint32_t*>(h.cardSet.ensure(64 + kCardSetSize * 4));
  HIP_OK(hipMemsetAsync(out, 0, 64, Runtime::get().stream));
  HIP_OK(hipMemsetAsync(out + 16, 0xff, kCardSetSize * 4, Runtime::get().stream));
  VX_LAUNCH("k_card_sample", k_card_sample, 16, 1024, 0, c, step, out + 16, out);
  uint32_t found = 0;
  copyOut(&found, VX355_MEM_HOST, out, 4);
  resetCounters(h);   // rows outside the sampled ranges touched the statistics
  h.sampledGroups = found;
}

// Rows [from, n) of the batch 'a' describes, in chunks.
void addInputGeneric(vx355_agg& h, AggArgs& a, int64_t n, int64_t from = 0) {
  h.mode = MODE_HASH;
  int64_t rows = 0;
  for (int64_t begin = from; begin < n; begin += rows) {
    // Every row of a chunk may be a new group: the chunk bounds the headroom the
    // group arrays need, so it grows with the table (4 M .. 64 M rows).
    const int64_t chunk = std::min<int64_t>(
        h.chunkRows, std::max<int64_t>(1 << 22, std::min<int64_t>(1 << 26, static_cast<int64_t>(h.gMaxGroups))));
    rows = std::min(chunk, n - begin);
    AggArgs c = a;
    shiftArgs(c, begin);
    runGeneric(h, c, rows, static_cast<uint64_t>(h.inputRows + begin), nullptr, false);
  }
}

// Keys stopped fitting a 64-bit normalized key in the middle of the stream (or
// a string key longer than 7 bytes showed up): move the live groups into the
// generic-mode structures and carry on there.
void switchToGeneric(vx355_agg& h) {
  auto& rt = Runtime::get();
  const uint64_t live = static_cast<uint64_t>(h.numGroups);
  const uint64_t newMax = std::max<uint64_t>(live + live / 2, 1024);
  h.gKeyStore.clear();
  h.gKeyStore.resize(h.keys.size());
  h.gCounter.ensure(64);
  HIP_OK(hipMemsetAsync(h.gCounter.ptr(), 0, 64, rt.stream));
  ToGenericArgs ta{};
  for (size_t k = 0; k < h.keys.size(); ++k) {
    const size_t w = static_cast<size_t>(keyStoreWords(h.keys[k].kind)) * 8;
    h.gKeyStore[k].ensure(newMax * w + 64);
    ta.g.keyStore[k] = h.gKeyStore[k].as<uint64_t>();
    ta.g.keyWords[k] = keyStoreWords(h.keys[k].kind);
    ta.range[k] = h.keys[k].range;
    ta.kind[k] = h.keys[k].kind;
  }
  h.gNullStore.ensure(newMax * 8 + 64);
  h.gHashStore.ensure(newMax * 8 + 64);
  DevBuf fresh;
  initTable(h, fresh, newMax);
  h.pairsComplete = false;
  if (h.tableReady && live > 0) {
    settleTable(h);
    ta.oldTable = h.table.as<uint64_t>();
    ta.oldRows = h.capacity;
    ta.oldMode = h.mode;
    ta.stride = h.stride;
    ta.numKeys = static_cast<int32_t>(h.keys.size());
    ta.newTable = fresh.as<uint64_t>();
    ta.g.nullStore = h.gNullStore.as<uint64_t>();
    ta.g.hashStore = h.gHashStore.as<uint64_t>();
    ta.g.gidCounter = h.gCounter.as<uint32_t>();
    VX_LAUNCH("k_to_generic", k_to_generic, streamGrid(static_cast<int64_t>(h.capacity), 256), 256, 0, ta);
    ++h.numRehashes;
    rt.sync();  // (k_rekey reads the old table, which the assignment below releases)
  }
  h.table = std::move(fresh);
  h.tableVirgin = false;
  h.capacity = newMax;
  h.gMaxGroups = newMax;
  h.tableReady = true;
  h.generic = true;
  h.mode = MODE_HASH;
  h.gSlotCap = 0;
  h.gSlots.release();
  ensureGenericCapacity(h, live);  // builds the slot array from the stored hashes
}

// Grid of the hi/lo split of every DOUBLE sum, chosen once per operator from the
// largest magnitude in a prefix of the first batch: values < 2^L, grid 2^(L-21),
// so up to 2^32 grid multiples add up exactly in the 53-bit significand of 'hi'
// whatever kernel, lane, workgroup or GPU adds them. A later value above 2^L is
// accumulated plainly (splitDouble): it only costs accuracy, never correctness.
void applySumStats(vx355_agg& h, AggArgs& a, const Counters& c);

// deferRead: the caller launches the key statistics behind this kernel and reads the counters once
// for both (applySumStats) - one stream synchronisation less on the first batch.
bool chooseSumGrids(vx355_agg& h, AggArgs& a, int64_t n, bool deferRead = false) {
  if (h.sumGridsChosen || !h.exactSums) {
    return false;
  }
  h.sumGridsChosen = true;
  bool any = false;
  for (int j = 0; j < a.numAccs; ++j) {
    any = any || a.accs[j].kind == ACC_SUM_F64;
  }
  if (!any) {
    return false;
  }
  if (deferRead) {
    return true;  // the caller runs k_first_stats: key statistics and these in one launch
  }
  AggArgs sa = a;
  sa.numRows = std::min<int64_t>(n, 1 << 16);
  resetCounters(h);
  // few workgroups: every wave ends with atomics on the same handful of counter words, and one
  // address retires < 100 M atomics per second
  VX_LAUNCH("k_sum_stats", k_sum_stats, std::min(streamGrid(sa.numRows, 256), 64), 256, 0, sa);
  applySumStats(h, a, readCounters(h));
  return false;
}

void applySumStats(vx355_agg& h, AggArgs& a, const Counters& c) {
  for (int j = 0; j < a.numAccs; ++j) {
    if (a.accs[j].kind != ACC_SUM_F64) {
      continue;
    }
    const uint64_t bits = c.sumMax[j];
    const int biased = static_cast<int>((bits >> 52) & 0x7ff);
    double m = 0;
    if (bits != 0 && biased != 0 && biased != 0x7ff) {
      const int L = (biased - 1023) + 1 + 4;
      const int G = L - 21;
      if (G + 52 < 1000 && G + 52 > -1000) {
        m = std::ldexp(1.5, G + 52);
      }
    }
    h.phys[a.accs[j].phys].splitM = m;
    a.accs[j].splitM = m;
  }
}

void addInput(vx355_agg& h, const vx355_batch* batch);

void flushPending(vx355_agg& h) {
  h.coalescer.flush([&](const vx355_batch* flat) { addInput(h, flat); });
}

bool tryCoalesce(vx355_agg& h, const vx355_batch* batch) {
  if (h.noMoreInput || !h.coalescer.append(batch, h.usedCols)) {
    return false;
  }
  if (h.coalescer.pendingRows() >= h.coalescer.thresholdRows) {
    flushPending(h);
  }
  return true;
}

void addInput(vx355_agg& h, const vx355_batch* batch) {
  auto& rt = Runtime::get();
  VX_CHECK_ARG(!h.noMoreInput, "addInput after noMoreInput");
  DeviceBatch db;
  db.load(batch, h.usedCols);
  const int64_t n = db.numRows();
  if (n == 0) {
    return;
  }
  ensureBasics(h);
  materializeAliases(h, db);

  AggArgs a{};
  a.numKeys = static_cast<int32_t>(h.keys.size());
  a.ignoreNullKeys = h.ignoreNullKeys ? 1 : 0;
  for (size_t k = 0; k < h.keys.size(); ++k) {
    a.keys[k].col = db.col(h.keys[k].col);
  }
  fillAccArgs(h, db, &a);
  patchAvgIntermediate(h, db, &a);
  if (h.maxProjRef >= static_cast<int32_t>(h.fusedProj.size())) {
    VX_THROW(VX355_EINVAL, "aggregate refers to a projection that vx355_agg_set_fused_input did not define");
  }
  a.numTerms = static_cast<int32_t>(h.fusedTerms.size());
  a.numProj = static_cast<int32_t>(h.fusedProj.size());
  makeTermArgs(db, h.fusedTerms.data(), a.numTerms, a.terms);
  makeProjectionArgs(db, h.fusedProj.data(), a.numProj, a.proj);
  a.stride = h.stride;
  a.counters = h.counters();

  const bool keyStatsFollow = !h.generic && !h.tableReady && a.numKeys > 0;
  const bool sumStatsPending = chooseSumGrids(h, a, n, keyStatsFollow);
  if (h.generic) {
    addInputGeneric(h, a, n);
    h.inputRows += n;
    rt.sync();
    return;
  }
  if (!h.tableReady) {
    // VectorHasher::analyze on a prefix of the first batch; later values that
    // fall outside are handled by the deferred-row path.
    resetCounters(h);
    bool needGeneric = false;
    bool statsPublished = false;
    if (a.numKeys > 0) {
      // Small first batches are analysed completely (no range widening later for
      // them); large ones by a 256 K-row prefix.
      const int64_t keyRows = n <= (8 << 20) ? n : (1 << 18);
      // (small prefixes only: a whole batch of up to 8 M rows wants the chip)
      // (128 blocks for a 256 K-row prefix: eight rows per thread, one dependent load after the other -
      // with 64 the pass took 17 us of BASELINE config 1's 150; every block ends with one pair of
      // atomics per key on the same two words, which is what keeps the number low)
      const int keyBlocks = keyRows <= (1 << 18) ? std::min(streamGrid(keyRows, 256), 128)
                                                 : std::min(streamGrid(keyRows, 256), rt.numCUs * 8);
      if (sumStatsPending) {
        AggArgs sa = a;
        sa.numRows = std::min<int64_t>(n, 1 << 16);
        // (every workgroup draws a ticket from ONE word: worth it for the small launches of a small
        // first batch only - with 2 K workgroups the tickets alone took longer than the read-back launch)
        const int statBlocks = keyBlocks + 1 + std::min(streamGrid(sa.numRows, 256), 64);
        statsPublished = statBlocks <= kMaxPublishingBlocks;
        VX_LAUNCH("k_first_stats", k_first_stats, statBlocks, 256, 0, sa, keyRows, keyBlocks,
                  statsPublished ? counterMail(h) : CounterMail{});
      } else {
        StatsArgs sa{};
        sa.numKeys = a.numKeys;
        for (int k = 0; k < a.numKeys; ++k) {
          sa.keys[k] = a.keys[k];
        }
        sa.numRows = keyRows;
        sa.counters = h.counters();
        VX_LAUNCH("k_key_stats", k_key_stats, keyBlocks + 1, 256, 0, sa);  // + the distinct probe's block
      }
      Counters c = statsPublished ? takePublishedCounters(h) : readAndResetCounters(h);
      h.firstRowsDistinct = c.firstRowsDistinct;
      if (sumStatsPending) {
        applySumStats(h, a, c);
      }
      needGeneric = c.unmappable != 0;  // a string key longer than 7 bytes
      c.unmappable = 0;
      checkCounters(c);
      mergeObserved(h, c);
    }
    if (!needGeneric) {
      try {
        // Sized for the first chunk (<= 1 M rows); the chunk loop grows an
        // open-addressing table before every later chunk (checkSize).
        rebuildTable(h, static_cast<uint64_t>(std::min<int64_t>({n, h.chunkRows, 1LL << 20})));
      } catch (const Error& e) {
        if (e.status != VX355_EUNSUPPORTED) {
          throw;
        }
        needGeneric = true;  // keys do not fit a 64-bit normalized key
      }
    }
    if (needGeneric) {
      // decideHashMode falls back to kHash (HashTable.cpp:1751-1839, cases 4/6).
      h.generic = true;
      addInputGeneric(h, a, n);
      h.inputRows += n;
      rt.sync();
      return;
    }
  }

  if (!h.cardSampled && ((h.mode == MODE_ARRAY && h.capacity > 8192) || h.mode == MODE_NORMALIZED) &&
      h.numGroups == 0 && n > 0 && !a.rowList) {
    sampleCardinality(h, a, n);
    if (h.mode == MODE_ARRAY && h.capacity > 8192 && h.sampledGroups >= 0 && h.sampledGroups < kCardSetSize / 2 - 64) {
      h.preferNormalized = true;
      rebuildTable(h, static_cast<uint64_t>(std::min<int64_t>({n, h.chunkRows, 1LL << 20})));
    }
  }
  int64_t rows = 0;
  for (int64_t begin = 0; begin < n; begin += rows) {
    // The first chunk of a stream is kept small: what it finds (number of live
    // groups) picks the LDS layout of every later launch.
    // (A direct-index table beyond the LDS path's 8192 groups has no layout to pick.)
    const bool wideArray = h.mode == MODE_ARRAY && h.capacity > 8192;
    // (nor has an open-addressing table in front of a large batch: the dense folds need no table)
    const bool fewGroups = h.sampledGroups >= 0 && std::max<int64_t>(h.numGroups, h.sampledGroups) <= kLdsHashedMaxGroups;
    const bool denseChunk = !fewGroups && radixDenseEligible(h, a, n - begin);   // few groups: the LDS kernels
    // (Nor does a small direct-index table whose first rows already show hundreds of keys: one LDS
    // word per key and accumulator will do, whatever the exact count - BASELINE config 1.)
    const bool manyKeysSeen = h.mode == MODE_ARRAY && h.capacity <= 8192 && h.firstRowsDistinct >= 256 &&
        h.inputRows == 0;
    rows = std::min((h.numGroups == 0 && !wideArray && !manyKeysSeen) ? std::min<int64_t>(h.chunkRows, 1 << 20)
                                                                      : h.chunkRows,
                    n - begin);
    if (denseChunk) {
      rows = std::min(n - begin, denseMaxRows(a));
    }
    if (wideArray && h.radixMinRows >= 0) {
      rows = std::min<int64_t>(rows, 1LL << radixRowBits(h.capacity));  // radix path: row numbers live in the records
    }
    // Few groups in an open-addressing table, aggregated by the LDS kernels with a hashed slot map:
    // the launch creates at most workgroups x slots groups (rows of a workgroup that runs out of
    // slots are deferred and replayed), so the table is sized for that and the chunk is not cut
    // at 2^26 rows - TPC-H Q1 with four keys: one 50 MB table and three launches instead of a
    // 12 GB table (2.5 ms to initialise, 1.9 ms to scan for live rows) and ten.
    uint64_t slotGroups = 0;
    if (h.mode == MODE_NORMALIZED && fewGroups && !denseChunk && h.slotsOnlyTables && h.numGroups > 0) {
      int words = 0;
      for (int j = 0; j < a.numAccs; ++j) {
        words += accWords(a.accs[j].kind);
      }
      LdsPlan probe{};
      size_t probeBytes = 0;
      if (chooseLds(h, words, &probe, &probeBytes)) {
        slotGroups = static_cast<uint64_t>(rt.numCUs) * 4 * static_cast<uint64_t>(probe.S);
      }
    }
    if (slotGroups > 0) {
      // (the deferred list holds every row of the chunk: 4 GB at most - 1 GB until round 6, which cut
      // TPC-H Q1 SF100 with four keys into four launches instead of two)
      rows = std::min<int64_t>(rows, 1LL << 30);
      if (h.tableDense || static_cast<uint64_t>(h.numGroups) + slotGroups > h.capacity * 7 / 10) {
        rebuildTable(h, slotGroups);
      }
    } else if (h.mode == MODE_NORMALIZED && !denseChunk) {
      // The open-addressing table is sized for the worst case "every row of the
      // chunk is a new group": keep that bound reasonable.
      rows = std::min<int64_t>(rows, 1 << 26);
      if (h.tableDense || static_cast<uint64_t>(h.numGroups + rows) > h.capacity * 7 / 10) {
        rebuildTable(h, static_cast<uint64_t>(rows));  // HashTable::checkSize (HashTable.cpp:772-806)
      }
    }
    const uint32_t deferCap =
        static_cast<uint32_t>(slotGroups > 0 ? rows : std::min<int64_t>(rows, h.deferCap));
    h.deferredBuf.ensure(static_cast<size_t>(deferCap) * 4 + 64);
    // Chunk = rows [begin, begin + rows): shift every column view instead of
    // adding an offset in the kernel.
    AggArgs c = a;
    c.numRows = rows;
    c.rowList = nullptr;
    c.rowBase = static_cast<uint64_t>(h.inputRows + begin);
    c.deferred = h.deferredBuf.as<int32_t>();
    c.deferCap = deferCap;
    auto shift = [&](ColView& v) {
      if (begin == 0 || v.enc == VX355_CONSTANT) {
        return;
      }
      // Chunks start on a multiple of 64 rows, so bitmaps shift by whole words.
      if (v.nulls) {
        v.nulls += begin >> 6;
      }
      if (v.enc == VX355_DICTIONARY) {
        v.indices += begin;
        return;
      }
      const int w = kindWidth(v.kind);
      if (w == 0) {
        v.values = static_cast<const uint64_t*>(v.values) + (begin >> 6);
      } else {
        v.values = static_cast<const char*>(v.values) + begin * w;
      }
    };
    for (int k = 0; k < c.numKeys; ++k) {
      shift(c.keys[k].col);
    }
    for (int j = 0; j < c.numAccs; ++j) {
      if (c.accs[j].hasIn) {
        shift(c.accs[j].in);
      }
      if (c.accs[j].hasMask) {
        shift(c.accs[j].mask);
      }
    }
    for (int t = 0; t < c.numTerms; ++t) {
      shift(c.terms[t].col);
    }
    for (int j = 0; j < c.numProj; ++j) {
      for (int f = 0; f < c.proj[j].numFactors; ++f) {
        if (c.proj[j].factors[f].hasCol) {
          shift(c.proj[j].factors[f].col);
        }
      }
    }
    int64_t pending = rows;
    const int32_t* list = nullptr;
    bool rescan = false;
    DevBuf replayList, oldRanges;
    for (int attempt = 0; pending > 0; ++attempt) {
      if (attempt > 64) {
        VX_THROW(VX355_EINTERNAL, "key range widening did not converge");
      }
      resetCounters(h);
      c.numRows = rescan ? rows : pending;
      c.rowList = rescan ? nullptr : list;
      c.rescanOld = rescan ? oldRanges.as<KeyRange>() : nullptr;
      c.table = h.table.as<uint64_t>();
      c.capacity = h.capacity;
      c.mode = h.mode;
      std::vector<KeyRange> used(c.numKeys);
      for (int k = 0; k < c.numKeys; ++k) {
        c.keys[k].range = h.keys[k].range;
        used[k] = h.keys[k].range;
      }
      h.denseNext = denseChunk && attempt == 0 && !rescan && list == nullptr && h.numGroups == 0;
      h.slotsOnlyLaunch = slotGroups > 0 && attempt == 0;
      h.countersPublished = false;
      launchChunk(h, c);
      h.denseNext = false;
      h.slotsOnlyLaunch = false;
      Counters ctr = h.countersPublished ? takePublishedCounters(h) : readAndResetCounters(h);
      h.countersPublished = false;
      // A key no VectorHasher range can hold (string longer than 7 bytes): its
      // rows were deferred; they force the generic mode below.
      bool toGeneric = ctr.unmappable != 0;
      ctr.unmappable = 0;
      checkCounters(ctr);
      if (h.pairsComplete && !ctr.pairsBroken) {
        h.pairCount += ctr.numNewGroups;  // a radix fold listed them (any other launch cleared pairsComplete)
      } else {
        h.pairsComplete = false;
      }
      h.numGroups += ctr.numNewGroups;
      pending = ctr.numDeferred;
      if (pending > 0) {
        // Widen, re-key, then replay the deferred rows; when the list
        // overflowed, rescan the chunk for the rows outside the ranges this
        // launch used.
        h.deferredRows += pending;
        mergeObserved(h, ctr);
        rescan = pending > static_cast<int64_t>(deferCap);
        if (rescan) {
          oldRanges.ensure(used.size() * sizeof(KeyRange) + 64);
          copyIn(oldRanges.ptr(), used.data(), VX355_MEM_HOST, used.size() * sizeof(KeyRange));
          rt.sync();
        } else {
          replayList.ensure(static_cast<size_t>(pending) * 4 + 64);
          copyIn(replayList.ptr(), h.deferredBuf.ptr(), VX355_MEM_DEVICE,
                 static_cast<size_t>(pending) * 4);
          list = replayList.as<int32_t>();
        }
        if (!toGeneric) {
          try {
            rebuildTable(h, static_cast<uint64_t>(pending));
          } catch (const Error& e) {
            if (e.status != VX355_EUNSUPPORTED) {
              throw;
            }
            toGeneric = true;  // the widened ranges no longer fit 64 bits
          }
        }
        if (toGeneric) {
          // decideHashMode falls back to kHash in the middle of the stream
          // (HashTable.cpp:1751-1839): convert the table, finish this chunk's
          // outstanding rows and the rest of the batch in generic mode.
          switchToGeneric(h);
          for (int k = 0; k < c.numKeys; ++k) {
            c.keys[k].range = used[k];
          }
          if (rescan) {
            runGeneric(h, c, rows, c.rowBase, nullptr, true);
          } else {
            for (int64_t at = 0; at < pending; at += 1 << 22) {
              runGeneric(h, c, std::min<int64_t>(1 << 22, pending - at), c.rowBase, list + at, false);
            }
          }
          addInputGeneric(h, a, n, begin + rows);
          h.inputRows += n;
          rt.sync();
          return;
        }
      }
    }
  }
  h.inputRows += n;
  rt.sync();
}

// First-seen order of 'n' listed entries {orderKeys: first row, orderVals: group row} by the packed
// two-level form (see k_fs_pack): h.order <- group rows by ascending first row; 'holes' of the entries have
// no group. false: not applicable (few entries, row numbers beyond 32 bits, switched off) - the
// caller sorts with the generic pair sort (radix_sort.hip).
bool sortFirstSeen(vx355_agg& h, size_t n, size_t holes) {
  auto& rt = Runtime::get();
  int64_t minEntries = 32LL << 20;   // measured at 10^8 entries; below this the generic sort's fixed costs are the smaller ones
  if (const char* e = std::getenv("VX355_AGG_OWN_SORT_MIN")) {
    minEntries = std::strtoll(e, nullptr, 10);  // < 0: never
  }
  // row numbers: those of the input, then - if some entries have no group - one per place in the list
  const uint64_t rows = static_cast<uint64_t>(std::max<int64_t>(1, h.inputRows)) + (holes ? n : 0);
  const int bits = std::max(21, 64 - __builtin_clzll(static_cast<unsigned long long>(rows)));
  if (minEntries < 0 || static_cast<int64_t>(n) < minEntries || bits > 20 + kFsMaxLowBits || !h.radixSorted) {
    return false;
  }
  uint64_t* packed = h.orderKeys.as<uint64_t>();
  uint64_t* tmp = static_cast<uint64_t*>(h.orderKeys2.ensure(n * 8 + 64));
  uint32_t* order = static_cast<uint32_t*>(h.orderVals2.ensure(n * 4 + 64));
  const uint32_t tileRecs = 65536;
  const int64_t tiles1 = ceilDiv(static_cast<int64_t>(n), tileRecs);
  const int64_t maxTiles2 = tiles1 + kSortBins;
  const int64_t cells1 = static_cast<int64_t>(kSortBins) * tiles1;
  const int64_t cells2 = static_cast<int64_t>(kSortBins) * maxTiles2;
  const int64_t parts = static_cast<int64_t>(kSortBins) * kSortBins;
  h.rpHist.ensure(static_cast<size_t>(std::max(cells1, cells2)) * 4 + 64);
  uint64_t* offsets1 = static_cast<uint64_t*>(h.rpOffsets.ensure(static_cast<size_t>(cells1 + 1 + cells2 + 1) * 8 + 64));
  uint64_t* offsets2 = offsets1 + cells1 + 1;
  h.rpTiles.ensure(static_cast<size_t>(std::max(tiles1 + 1, maxTiles2)) * sizeof(RadixTile) + 64);
  // [0..1] tile counts, [4] error flag, [8..11] the bounds of the one
  // "bucket" of level 1, then the two partition -> cell maps
  uint32_t* misc = static_cast<uint32_t*>(h.rpMisc.ensure(64 + static_cast<size_t>(kSortBins + 1 + parts + 1) * 4 + 64));
  uint32_t* partCell1 = misc + 16;
  uint32_t* partCell2 = partCell1 + kSortBins + 1;
  const uint64_t bounds[2] = {0, static_cast<uint64_t>(n)};
  HIP_OK(hipMemsetAsync(misc, 0, 64, rt.stream));
  copyIn(misc + 8, bounds, VX355_MEM_HOST, 16);
  VX_LAUNCH("k_fs_pack", k_fs_pack, streamGrid(static_cast<int64_t>(n), 256), 256, 0, packed, h.orderVals.as<uint32_t>(),
            static_cast<uint64_t>(n), static_cast<uint32_t>(std::max<int64_t>(1, h.inputRows)));
  Radix2Args r{};
  r.tiles = h.rpTiles.as<RadixTile>();
  r.recWords = 1;
  r.numBins = kSortBins;
  r.hist = h.rpHist.as<uint32_t>();
  auto level = [&](const Level1Bins& from, int32_t numBins1, int64_t numParts, int64_t cells, uint32_t* numTiles,
                   uint32_t* partCell, const uint64_t* in, uint64_t* out, uint64_t* offsets, int32_t shiftB, int64_t tiles) {
    VX_LAUNCH("k_fs_tiles", k_rp_tiles, static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(64, numParts >> 14))), 1024, 0,
              from, numBins1, kSortBins, 10, tileRecs, h.rpTiles.as<RadixTile>(),
              numTiles, partCell, numParts);
    r.in = in;
    r.out = out;
    r.numTiles = numTiles;
    r.shiftB = shiftB;
    r.offsets = offsets;
    HIP_OK(hipMemsetAsync(r.hist, 0, static_cast<size_t>(cells) * 4, rt.stream));
    const int grid = static_cast<int>(std::min<int64_t>(tiles, rt.numCUs * 2));
    VX_LAUNCH("k_fs_count", k_rp_count2, grid, 1024, 0, r);
    scanU32ToU64(r.hist, cells, offsets, h.rpScan);
    VX_LAUNCH("k_fs_scatter", k_fs_scatter, grid, kSortThreads, 0, r);
  };
  Level1Bins whole{};
  whole.offsets1 = reinterpret_cast<const uint64_t*>(misc + 8);
  whole.numTiles1 = 1;
  level(whole, 1, kSortBins, cells1, misc, partCell1, packed, tmp, offsets1, bits - 10, tiles1);
  Level1Bins bins1{};
  bins1.offsets1 = offsets1;
  bins1.numTiles1 = tiles1;
  level(bins1, kSortBins, parts, cells2, misc + 1, partCell2, tmp, packed, offsets2, bits - 20, maxTiles2);
  VX_LAUNCH("k_fs_rank", k_fs_rank, rt.numCUs * 8, 256, 0, packed, offsets2, partCell2, parts, bits - 20, order, misc + 4);
  uint32_t error = 0;
  copyOut(&error, VX355_MEM_HOST, misc + 4, 4);
  if (error != 0) {
    VX_THROW(VX355_EINTERNAL, "first-seen sort: two entries share a first input row");
  }
  h.order = order;
  return true;
}

void finalize(vx355_agg& h) {
  auto& rt = Runtime::get();
  ensureBasics(h);
  if (h.keys.empty()) {
    // Global aggregation: exactly one output row, even without input
    // (GroupingSet.cpp:623-669).
    if (!h.tableReady) {
      rebuildTable(h, 0);
    }
    h.numOutput = 1;
    return;
  }
  if (!h.tableReady || h.numGroups == 0) {
    h.numOutput = 0;
    return;
  }
  const size_t g = static_cast<size_t>(h.numGroups);
  settleTable(h);
  // Every group was listed by the radix folds that created it: no scan of the table.
  // (a list with holes - dense folds - is only good for sorting: the holes go behind the groups)
  const bool listed = h.pairsComplete && h.pairCount == h.numGroups && !(h.unorderedOutput && h.pairHoles != 0);
  const size_t listLen = listed ? g + static_cast<size_t>(h.pairHoles) : g;
  if (!listed && !h.unorderedOutput && h.capacity <= 65536 && g <= static_cast<size_t>(kSmallSortMax)) {
    uint32_t* order = static_cast<uint32_t*>(h.orderVals.ensure(g * 4 + 64));
    VX_LAUNCH("k_collect_sort_small", k_collect_sort_small, static_cast<int>(ceilDiv(static_cast<int64_t>(g), 64)), 1024,
              0, h.table.as<uint64_t>(), static_cast<uint32_t>(h.capacity), h.stride, order, static_cast<uint32_t>(g),
              order + g);
    h.order = order;
    h.numOutput = static_cast<int64_t>(g);
    h.collectCheck = static_cast<int64_t>(g);  // verified behind the first output page's synchronisation
    return;
  }
  if (!listed) {
    h.orderKeys.ensure(g * 8 + 64);
    h.orderVals.ensure(g * 4 + 64);
    uint32_t* cursor = reinterpret_cast<uint32_t*>(h.counters());
    resetCounters(h);
    VX_LAUNCH("k_collect", k_collect, streamGrid(static_cast<int64_t>(h.capacity), 256), 256, 0,
              h.table.as<uint64_t>(), h.capacity, h.stride, h.orderKeys.as<uint64_t>(),
              h.orderVals.as<uint32_t>(), cursor);
    if (!h.unorderedOutput && g <= static_cast<size_t>(kSmallSortMax)) {
      // few groups in a large table (TPC-H Q1 with four keys: 196 of them): ranked by one workgroup
      // straight from the list, no read-back of the count, no device-wide radix sort
      uint32_t* order = static_cast<uint32_t*>(h.orderVals2.ensure(g * 4 + 64));
      VX_LAUNCH("k_rank_sort_small", k_rank_sort_small, static_cast<int>(ceilDiv(static_cast<int64_t>(g), 64)), 1024, 0,
                h.orderKeys.as<uint64_t>(),
                h.orderVals.as<uint32_t>(), cursor, order, static_cast<uint32_t>(g));
      copyIn(order + g, cursor, VX355_MEM_DEVICE, 4);  // the count, for the check behind the output page
      h.order = order;
      h.numOutput = static_cast<int64_t>(g);
      h.collectCheck = static_cast<int64_t>(g);
      return;
    }
    Counters c = readCounters(h);
    const uint32_t found = c.numDeferred;  // first word of the block is the cursor
    if (found != g) {
      VX_THROW(VX355_EINTERNAL, "group count mismatch: counted " + std::to_string(g) + ", found " +
                                    std::to_string(found));
    }
  }
  h.orderKeys2.ensure(listLen * 8 + 64);
  h.orderVals2.ensure(listLen * 4 + 64);
  if (h.unorderedOutput) {
    rt.sync();
    h.order = h.orderVals.as<uint32_t>();  // table order as k_collect found it
    h.numOutput = static_cast<int64_t>(g);
    return;
  }
  // first rows are < inputRows: sort only the bits that can be set (a hole has all of them set, and
  // inputRows - 1 has not)
  const int bits = std::max(1, 64 - __builtin_clzll(static_cast<unsigned long long>(std::max<int64_t>(1, h.inputRows))));
  if (sortFirstSeen(h, listLen, listLen - g)) {
    h.numOutput = static_cast<int64_t>(g);
    return;
  }
  bool inTmp = false;
  sortPairsU64U32(h.orderKeys.as<uint64_t>(), h.orderVals.as<uint32_t>(), h.orderKeys2.as<uint64_t>(),
                  h.orderVals2.as<uint32_t>(), listLen, h.sortTmp, &inTmp, bits);
  rt.sync();
  h.order = inTmp ? h.orderVals2.as<uint32_t>() : h.orderVals.as<uint32_t>();
  h.numOutput = static_cast<int64_t>(g);
}

// HBM held by the group table (+ the set tables of DISTINCT aggregates / min / max over strings):
// what GroupingSet::isPartialFull compares with max_partial_aggregation_memory.
static int64_t tableBytesOf(const vx355_agg& h) {
  int64_t bytes = static_cast<int64_t>(h.table.capacity());
  for (const auto& d : h.distinct) {
    if (d.dedup) {
      bytes += static_cast<int64_t>(d.dedup->table.capacity());
      for (const auto& b : d.dedup->strBlocks) {
        bytes += static_cast<int64_t>(b.capacity());
      }
    }
  }
  return bytes;
}

static void publishStats(vx355_agg& h) {
  h.publishedTableBytes.store(tableBytesOf(h), std::memory_order_relaxed);
  h.publishedGroups.store(h.keys.empty() ? 1 : h.numGroups, std::memory_order_relaxed);
  // bytes the groups held now occupy: the allocation scaled by the share of its rows in use (the
  // reference's measure after a flush - resetTable keeps the table, the rows are gone,
  // exec/HashAggregation.cpp:293-318) plus the string blocks of the DISTINCT sets
  const int64_t groups = h.keys.empty() ? 1 : h.numGroups;
  const int64_t rows = std::max<int64_t>(1, static_cast<int64_t>(h.capacity));
  int64_t used = static_cast<int64_t>(static_cast<double>(h.table.capacity()) * std::min<double>(1.0, static_cast<double>(groups) / rows));
  for (const auto& d : h.distinct) {
    if (d.dedup) {
      used += static_cast<int64_t>(d.dedup->table.capacity());
      for (const auto& b : d.dedup->strBlocks) {
        used += static_cast<int64_t>(b.capacity());
      }
    }
  }
  h.publishedUsedBytes.store(used, std::memory_order_relaxed);
}

// GroupingSet::resetTable after a partial flush (HashAggregation::resetPartialOutputIfNeed,
// HashAggregation.cpp:293-318): the table is emptied, key ranges and the mode stay.
void resetAfterFlush(vx355_agg& h) {
  auto& rt = Runtime::get();
  if (h.generic) {
    h.strBlocks.clear();  // the flushed pages were drained: their strings are no longer referenced
    h.strCap = h.strUsedHost = 0;
    if (h.gSlotCap) {
      HIP_OK(hipMemsetAsync(h.gSlots.ptr(), 0, static_cast<size_t>(h.gSlotCap) * 8, rt.stream));
    }
    if (h.gCounter.ptr()) {
      HIP_OK(hipMemsetAsync(h.gCounter.ptr(), 0, 64, rt.stream));
    }
  }
  if (h.tableReady && h.tableDense) {
    h.numGroups = 0;
    rebuildTable(h, 0);  // an empty open-addressing table in place of the array of rows
  } else if (h.tableReady) {
    initTable(h, h.table, h.capacity);
    h.tableVirgin = false;
  }
  rt.sync();
  h.pairCount = 0;
  h.pairHoles = 0;
  h.pairsComplete = true;
  h.numGroups = 0;
  h.numOutput = -1;
  h.outputCursor = 0;
  h.order = nullptr;
  h.flushing = false;
  ++h.numFlushes;
  publishStats(h);  // vx355_agg_table_bytes: the table is empty again
}

// ---- min / max over VARCHAR / VARBINARY ----------------------------------------------------
// (MinMaxAggregateBase.cpp:305-480, SingleValueAccumulator.) The distinct (keys, string) pairs
// of the operator live in a dedup table (see vx355_agg::DistinctPart); at noMoreInput their
// strings are ranked by an LSD radix sort over 8-byte big-endian words - length first (it only
// decides between a string and its zero-padded extension), then the words from the last to
// the first - and the aggregation becomes min / max over BIGINT ranks.

// Bytes [8 * level, 8 * level + 8) of the string, big-endian, zero-padded.
__device__ inline uint64_t stringWordBE(const uint4 v, int level) {
  const uint32_t size = v.x;
  const uint32_t off = 8u * static_cast<uint32_t>(level);
  if (off >= size) {
    return 0;
  }
  uint64_t r = 0;
  if (size <= 12) {
    const uint32_t w[3] = {v.y, v.z, v.w};
    for (uint32_t b = 0; b < 8 && off + b < size; ++b) {
      const uint32_t i = off + b;
      r |= static_cast<uint64_t>((w[i >> 2] >> ((i & 3) * 8)) & 0xff) << (56 - 8 * b);
    }
  } else {
    const unsigned char* p =
        reinterpret_cast<const unsigned char*>((static_cast<uint64_t>(v.w) << 32) | v.z);
    for (uint32_t b = 0; b < 8 && off + b < size; ++b) {
      r |= static_cast<uint64_t>(p[off + b]) << (56 - 8 * b);
    }
  }
  return r;
}

__global__ __launch_bounds__(256) void k_str_max_len(const uint4* views, const uint64_t* nulls, int64_t n,
                                                     uint32_t* out) {
  uint32_t m = 0;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    if (!nulls || ((nulls[i >> 6] >> (i & 63)) & 1)) {
      m = max(m, views[i].x);
    }
  }
  for (int d = kWave / 2; d > 0; d >>= 1) {
    m = max(m, static_cast<uint32_t>(__shfl_xor(static_cast<int>(m), d, kWave)));
  }
  if (lane() == 0 && m) {
    atomicMax(out, m);
  }
}

// level < 0: the length
__global__ __launch_bounds__(256) void k_str_sort_key(const uint4* views, const uint64_t* nulls,
                                                      const uint32_t* perm, int64_t n, int level, uint64_t* keys) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) {
    return;
  }
  const uint32_t row = perm[i];
  uint64_t k = 0;
  if (!nulls || ((nulls[row >> 6] >> (row & 63)) & 1)) {
    const uint4 v = views[row];
    k = level < 0 ? v.x : stringWordBE(v, level);
  }
  keys[i] = k;
}

__global__ __launch_bounds__(256) void k_iota_u32(uint32_t* out, int64_t n) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < n) {
    out[i] = static_cast<uint32_t>(i);
  }
}

__global__ __launch_bounds__(256) void k_rank_from_perm(const uint32_t* perm, int64_t n, int64_t* rank) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < n) {
    rank[perm[i]] = i;
  }
}

// The aggregate's output column: rank -> the string that holds it.
__global__ __launch_bounds__(256) void k_rank_to_view(const int64_t* ranks, const uint64_t* rankNulls, int32_t n,
                                                      const uint32_t* perm, const uint4* views, uint4* out,
                                                      uint64_t* outNulls) {
  const int32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos - static_cast<int32_t>(lane()) >= n) {
    return;
  }
  const bool valid = pos < n && ((rankNulls[pos >> 6] >> (pos & 63)) & 1);
  writeBit(outNulls, pos, valid);
  if (pos < n) {
    out[pos] = valid ? views[perm[ranks[pos]]] : make_uint4(0, 0, 0, 0);
  }
}

// toIntermediate of min / max over strings: the value itself where the mask lets the row through
// (MinMaxAggregateBase.cpp:319-349).
__global__ __launch_bounds__(256) void k_string_pass(ColView in, ColView mask, int32_t hasMask, int32_t n, uint4* out,
                                                     uint64_t* outNulls) {
  const int32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos - static_cast<int32_t>(lane()) >= n) {
    return;
  }
  bool valid = pos < n && !colIsNull(in, pos);
  if (valid && hasMask) {
    valid = !colIsNull(mask, pos) && loadInt64(mask, colIndex(mask, pos)) != 0;
  }
  writeBit(outNulls, pos, valid);
  if (pos < n) {
    out[pos] = valid ? static_cast<const uint4*>(in.values)[colIndex(in, pos)] : make_uint4(0, 0, 0, 0);
  }
}

constexpr size_t kSmallPageBytes = 1 << 20;
constexpr size_t kDirectPageBytes = 256 << 10;  // pages k_extract writes into pinned host memory itself

void getOutput(vx355_agg& h, vx355_out_column* cols, int32_t numCols, int32_t maxRows, int32_t* nOut,
               int32_t* finished) {
  auto& rt = Runtime::get();
  VX_CHECK_ARG((cols || numCols == 0) && nOut && finished, "NULL argument");
  VX_CHECK_ARG(numCols == static_cast<int32_t>(h.outTypes.size()), "wrong number of output columns");
  VX_CHECK_ARG(maxRows > 0, "max_rows must be positive");
  VX_CHECK_ARG(h.noMoreInput || h.flushing, "getOutput before noMoreInput (vx355_agg_flush opens a partial flush)");
  if (h.numOutput < 0) {
    finalize(h);
  }
  const int32_t n = static_cast<int32_t>(std::min<int64_t>(maxRows, h.numOutput - h.outputCursor));
  *nOut = n;
  if (n <= 0) {
    *nOut = 0;
    *finished = 1;
    if (h.flushing) {
      resetAfterFlush(h);
    }
    return;
  }
  for (int32_t c = 0; c < numCols; ++c) {
    VX_CHECK_ARG(cols[c].type_kind == h.outTypes[c], "output column type mismatch");
    const bool skippedKey = h.keysOptional && c < static_cast<int32_t>(h.keys.size()) &&
        cols[c].mem == VX355_MEM_DEVICE;
    VX_CHECK_ARG(cols[c].values != nullptr || skippedKey, "output column without values buffer");
  }
  // Device scratch for host-resident output columns.
  const size_t words = static_cast<size_t>(ceilDiv(n, 64));
  std::vector<size_t> valueBytes(numCols), offsets(numCols), nullOffsets(numCols);
  size_t total = 0;
  for (int32_t c = 0; c < numCols; ++c) {
    const int w = kindWidth(cols[c].type_kind);
    valueBytes[c] = w == 0 ? words * 8 : static_cast<size_t>(n) * w;
    offsets[c] = total;
    total += (valueBytes[c] + 63) & ~static_cast<size_t>(63);
    nullOffsets[c] = total;
    total += (words * 8 + 63) & ~static_cast<size_t>(63);
  }
  bool anyHost = false;
  for (int32_t i = 0; i < numCols; ++i) {
    anyHost = anyHost || cols[i].mem == VX355_MEM_HOST;
  }
  // A small page for host columns (BASELINE config 1: 1000 rows of three columns, Q1: four rows of
  // ten) is written by k_extract straight into the handle's pinned, device-mapped stage: no scratch
  // block in HBM and no copy command behind the kernel (round 6; one launch less per page).
  const bool directStage = anyHost && total <= kDirectPageBytes;
  char* stage = nullptr;
  char* scratch = nullptr;
  if (directStage) {
    h.outStage.clear();
    stage = h.outStage.extend(total + 64);
    HIP_OK(hipHostGetDevicePointer(reinterpret_cast<void**>(&scratch), stage, 0));
  } else {
    scratch = static_cast<char*>(h.scratch.ensure(total + 64));
  }
  auto devValues = [&](int32_t c) -> void* {
    return cols[c].mem == VX355_MEM_HOST ? static_cast<void*>(scratch + offsets[c]) : cols[c].values;
  };
  auto devNulls = [&](int32_t c) -> uint64_t* {
    if (cols[c].mem == VX355_MEM_HOST) {
      return reinterpret_cast<uint64_t*>(scratch + nullOffsets[c]);
    }
    return cols[c].nulls;
  };
  ExtractArgs ea{};
  ea.table = h.table.as<uint64_t>();
  ea.stride = h.stride;
  ea.mode = h.mode;
  ea.order = h.order;
  ea.begin = h.outputCursor;
  ea.count = n;
  ea.numKeys = static_cast<int32_t>(h.keys.size());
  ea.numAggs = static_cast<int32_t>(h.aggs.size());
  ea.global = h.keys.empty() ? 1 : 0;
  int32_t c = 0;
  for (size_t k = 0; k < h.keys.size(); ++k, ++c) {
    ea.keys[k].values = devValues(c);
    ea.keys[k].nulls = devNulls(c);
    ea.keys[k].kind = h.keys[k].kind;
    ea.keys[k].range = h.keys[k].range;
    ea.keys[k].keyIndex = static_cast<int32_t>(k);
    if (h.generic) {
      ea.keys[k].store = h.gKeyStore[k].as<uint64_t>();
      ea.keys[k].storeWords = keyStoreWords(h.keys[k].kind);
    }
  }
  ea.nullStore = h.generic ? h.gNullStore.as<uint64_t>() : nullptr;
  auto physOff = [&](int32_t p) {
    if (p < 0) {
      return -1;
    }
    while (h.phys[p].aliasOf >= 0) {
      p = h.phys[p].aliasOf;
    }
    return flagFromFirstRow(h, p) ? 1 : offOf(h, p);
  };
  const bool fin = finalOutput(h.step);
  for (size_t j = 0; j < h.aggs.size(); ++j) {
    const auto& la = h.aggs[j];
    OutAgg& oa = ea.aggs[j];
    oa.aggKind = la.fn.kind;
    oa.inputType = la.fn.input_type;
    oa.mainOff = physOff(la.main);
    oa.loOff = (la.main >= 0 && accWords(h.phys[la.main].kind) == 2) ? oa.mainOff + 1 : -1;
    oa.seenOff = physOff(la.seen);
    oa.finalOut = fin ? 1 : 0;
    oa.values = devValues(c);
    oa.nulls = devNulls(c);
    ++c;
    if (la.fn.kind == VX355_AGG_AVG && !fin) {
      oa.values2 = devValues(c);
      oa.nulls2 = devNulls(c);
      ++c;
    }
  }
  ea.overflow = &h.counters()->overflow;
  bool checksTotals = false;
  for (int j = 0; j < ea.numAggs; ++j) {
    checksTotals = checksTotals || (ea.aggs[j].aggKind == VX355_AGG_SUM && ea.aggs[j].inputType <= VX355_BIGINT &&
                                    ea.aggs[j].loOff >= 0);
  }
  if (checksTotals) {
    resetCounters(h);
  }
  const int64_t expectFound = h.collectCheck;
  if (expectFound >= 0) {
    h.collectCheck = -1;
    ea.checkSrc = h.order + expectFound;
    ea.checkDst = reinterpret_cast<uint32_t*>(rt.mail.dev + 63);
  }
  VX_LAUNCH("k_extract", k_extract, static_cast<int>(ceilDiv(n, 256)), 256, 0, ea);
  if (checksTotals) {
    checkCounters(readCounters(h));  // "integer overflow": a sum(BIGINT) total left int64
  }
  struct FoundCheck {
    vx::Runtime& rt;
    int64_t expect;
    void operator()() const {
      if (expect >= 0 && static_cast<int64_t>(static_cast<uint32_t>(rt.mail.host[63])) != expect) {
        VX_THROW(VX355_EINTERNAL, "group count mismatch: counted " + std::to_string(expect) + ", found " +
                                      std::to_string(static_cast<uint32_t>(rt.mail.host[63])));
      }
    }
  } foundCheck{rt, expectFound};
  if (anyHost && total <= kSmallPageBytes) {
    // A small page: one copy of the whole scratch block into pinned memory instead of twenty copies
    // into the caller's pageable buffers (each of those is a separate synchronous transfer: 0.4 ms
    // for Q1's page, 0.06 ms this way) - or no copy at all (directStage).
    if (!directStage) {
      h.outStage.clear();
      stage = h.outStage.extend(total);
      copyOutAsync(stage, VX355_MEM_HOST, scratch, total);
    } else {
      rt.d2hBytes += total;  // (the kernel's own stores crossed the link: the same bytes a copy would have moved)
    }
    rt.sync();
    foundCheck();
    for (int32_t i = 0; i < numCols; ++i) {
      if (cols[i].mem == VX355_MEM_HOST) {
        std::memcpy(cols[i].values, stage + offsets[i], valueBytes[i]);
        if (cols[i].nulls) {
          std::memcpy(cols[i].nulls, stage + nullOffsets[i], words * 8);
        }
      }
    }
  } else {
    for (int32_t i = 0; i < numCols; ++i) {
      if (cols[i].mem == VX355_MEM_HOST) {
        copyOutAsync(cols[i].values, VX355_MEM_HOST, scratch + offsets[i], valueBytes[i]);
        if (cols[i].nulls) {
          copyOutAsync(cols[i].nulls, VX355_MEM_HOST, scratch + nullOffsets[i], words * 8);
        }
      }
    }
    rt.sync();
    foundCheck();
  }
  if (h.generic && h.hasStringKeys) {
    // Key strings longer than 12 bytes came out as views into the operator's HBM arena. Device
    // output columns keep those (valid while the handle lives); for host columns the bytes follow
    // into a buffer owned by the handle (valid until its next get_output) and the views are
    // re-pointed.
    h.hostStrings.clear();
    for (size_t k = 0; k < h.keys.size(); ++k) {
      if (isString(h.keys[k].kind) && cols[k].mem == VX355_MEM_HOST) {
        fetchLongStrings(static_cast<char*>(cols[k].values), n, h.hostStrings);
      }
    }
  }
  h.outputCursor += n;
  *finished = h.outputCursor >= h.numOutput ? 1 : 0;
  if (*finished && h.flushing) {
    resetAfterFlush(h);
  }
}

// GroupingSet::toIntermediate for one batch: see include/vx355.h.
void toIntermediate(vx355_agg& h, const vx355_batch* batch, vx355_out_column* cols, int32_t numCols) {
  auto& rt = Runtime::get();
  VX_CHECK_ARG(batch && (cols || numCols == 0), "NULL argument");
  if (!rawInput(h.step)) {
    VX_THROW(VX355_EINVAL, "toIntermediate applies to raw input (partial / single steps); intermediate input passes through");
  }
  if (!h.fusedTerms.empty() || !h.fusedProj.empty()) {
    VX_THROW(VX355_EUNSUPPORTED, "toIntermediate with a fused FilterProject");
  }
  const int32_t expected = static_cast<int32_t>(h.outTypes.size() - h.keys.size());
  VX_CHECK_ARG(numCols == expected, "wrong number of aggregate output columns");
  std::vector<int32_t> used;
  for (const auto& la : h.aggs) {
    used.push_back(la.fn.input_col);
    used.push_back(la.fn.mask_col);
  }
  DeviceBatch db;
  db.load(batch, used);
  const int32_t n = db.numRows();
  if (n == 0) {
    return;
  }
  // the PARTIAL layout of the aggregate columns, whatever this operator's own step
  std::vector<int32_t> types;
  for (const auto& la : h.aggs) {
    const auto& f = la.fn;
    const bool isInt = isIntLike(f.input_type);
    switch (f.kind) {
      case VX355_AGG_SUM:
        types.push_back(isInt ? VX355_BIGINT : VX355_DOUBLE);
        break;
      case VX355_AGG_COUNT:
      case VX355_AGG_COUNT_STAR:
        types.push_back(VX355_BIGINT);
        break;
      case VX355_AGG_MIN:
      case VX355_AGG_MAX:
        types.push_back(f.input_type);
        break;
      default:
        types.push_back(VX355_DOUBLE);
        types.push_back(VX355_BIGINT);
        break;
    }
  }
  // (a SINGLE / FINAL operator has one avg column in outTypes; the intermediate layout has two)
  VX_CHECK_ARG(h.step == VX355_STEP_PARTIAL || types.size() >= static_cast<size_t>(numCols), "column count");
  VX_CHECK_ARG(static_cast<int32_t>(types.size()) == numCols || h.step != VX355_STEP_PARTIAL,
               "wrong number of aggregate output columns");
  if (static_cast<int32_t>(types.size()) != numCols) {
    VX_THROW(VX355_EINVAL, "toIntermediate needs the PARTIAL step's column layout (two columns per avg)");
  }
  const size_t words = static_cast<size_t>(ceilDiv(n, 64));
  std::vector<size_t> valueBytes(numCols), offsets(numCols), nullOffsets(numCols);
  size_t total = 0;
  for (int32_t c = 0; c < numCols; ++c) {
    VX_CHECK_ARG(cols[c].type_kind == types[c], "output column type mismatch");
    VX_CHECK_ARG(cols[c].values != nullptr, "output column without values buffer");
    const int w = kindWidth(types[c]);
    valueBytes[c] = w == 0 ? words * 8 : static_cast<size_t>(n) * w;
    offsets[c] = total;
    total += (valueBytes[c] + 63) & ~static_cast<size_t>(63);
    nullOffsets[c] = total;
    total += (words * 8 + 63) & ~static_cast<size_t>(63);
  }
  bool anyHost = false;
  for (int32_t i = 0; i < numCols; ++i) {
    anyHost = anyHost || cols[i].mem == VX355_MEM_HOST;
  }
  // A small page for host columns (BASELINE config 1: 1000 rows of three columns, Q1: four rows of
  // ten) is written by k_extract straight into the handle's pinned, device-mapped stage: no scratch
  // block in HBM and no copy command behind the kernel (round 6; one launch less per page).
  const bool directStage = anyHost && total <= kDirectPageBytes;
  char* stage = nullptr;
  char* scratch = nullptr;
  if (directStage) {
    h.outStage.clear();
    stage = h.outStage.extend(total + 64);
    HIP_OK(hipHostGetDevicePointer(reinterpret_cast<void**>(&scratch), stage, 0));
  } else {
    scratch = static_cast<char*>(h.scratch.ensure(total + 64));
  }
  auto devValues = [&](int32_t c) -> void* {
    return cols[c].mem == VX355_MEM_HOST ? static_cast<void*>(scratch + offsets[c]) : cols[c].values;
  };
  auto devNulls = [&](int32_t c) -> uint64_t* {
    return cols[c].mem == VX355_MEM_HOST ? reinterpret_cast<uint64_t*>(scratch + nullOffsets[c]) : cols[c].nulls;
  };
  ToIntermediateArgs ta{};
  ta.numAggs = static_cast<int32_t>(h.aggs.size());
  ta.count = n;
  int32_t c = 0;
  for (size_t j = 0; j < h.aggs.size(); ++j) {
    const auto& f = h.aggs[j].fn;
    ToIntermediateAgg& g = ta.aggs[j];
    g.aggKind = f.kind;
    g.inputType = f.input_type;
    g.inIsInt = isIntLike(f.input_type) ? 1 : 0;
    g.hasIn = f.kind == VX355_AGG_COUNT_STAR ? 0 : 1;
    if (g.hasIn) {
      g.in = db.col(f.input_col);
    }
    g.hasMask = f.mask_col >= 0 ? 1 : 0;
    if (g.hasMask) {
      g.mask = db.col(f.mask_col);
    }
    g.values = devValues(c);
    g.nulls = devNulls(c);
    ++c;
    if (f.kind == VX355_AGG_AVG) {
      g.values2 = devValues(c);
      g.nulls2 = devNulls(c);
      ++c;
    }
  }
  VX_LAUNCH("k_to_intermediate", k_to_intermediate, static_cast<int>(ceilDiv(n, 256)), 256, 0, ta);
  for (int32_t i = 0; i < numCols; ++i) {
    if (cols[i].mem == VX355_MEM_HOST) {
      copyOutAsync(cols[i].values, VX355_MEM_HOST, scratch + offsets[i], valueBytes[i]);
      if (cols[i].nulls) {
        copyOutAsync(cols[i].nulls, VX355_MEM_HOST, scratch + nullOffsets[i], words * 8);
      }
    }
  }
  rt.sync();
}

// Knobs every operator (and the dedup / outer operators of a DISTINCT aggregate) reads.
void configureFromEnv(vx355_agg& h) {
  if (const char* e = std::getenv("VX355_ARRAY_MAX")) {
    h.arrayMax = std::strtoull(e, nullptr, 10);
  }
  if (const char* e = std::getenv("VX355_JIT")) {
    // 0 = off; 1 / sync = compile on the calling thread; async (the default) = on a helper thread
    // while the interpreting kernel takes the first batches
    h.jitEnabled = e[0] != '0';
    h.jitAsync = !(std::string(e) == "1" || std::string(e) == "sync");
  }
  if (const char* e = std::getenv("VX355_AGG_COALESCE_ROWS")) {
    h.coalescer.thresholdRows = std::strtoll(e, nullptr, 10);
  }
  if (const char* e = std::getenv("VX355_EXACT_SUMS")) {
    h.exactSums = e[0] != '0';
  }
  if (const char* e = std::getenv("VX355_LOG_SHAPES")) {
    h.logShapes = e[0] == '1';
  }
  if (const char* e = std::getenv("VX355_AGG_DEFER_CAP")) {
    h.deferCap = std::max<int64_t>(1, std::strtoll(e, nullptr, 10));
  }
  if (const char* e = std::getenv("VX355_AGG_LDS_BLOCKS_PER_CU")) {
    h.ldsBlocksPerCuCap = std::atoi(e);
  }
  if (const char* e = std::getenv("VX355_AGG_FAST_UNROLL")) {
    h.fastUnroll = std::atoi(e) == 2 ? 2 : 4;
  }
  if (const char* e = std::getenv("VX355_AGG_NO_FAST")) {
    h.disableFast = e[0] == '1';
  }
  if (const char* e = std::getenv("VX355_AGG_SLOTS_ONLY")) {
    h.slotsOnlyTables = std::atoi(e) != 0;
  }
  if (const char* e = std::getenv("VX355_AGG_SCRATCH_FLUSH")) {
    h.ldsScratchFlush = std::atoi(e) != 0;
  }
  if (const char* e = std::getenv("VX355_AGG_SCRATCH_BLOCKS_PER_CU")) {
    h.scratchBlocksPerCu = std::max(1, std::atoi(e));
  }
  if (const char* e = std::getenv("VX355_AGG_SCRATCH_MIN_ATOMICS")) {
    h.scratchMinAtomics = std::strtoll(e, nullptr, 10);
  }
  if (const char* e = std::getenv("VX355_AGG_RADIX_SPARSE")) {
    h.radixSparse = std::atoi(e) != 0;
  }
  if (const char* e = std::getenv("VX355_AGG_RADIX_DENSE")) {
    h.radixDense = std::atoi(e) != 0;
  }
  if (const char* e = std::getenv("VX355_AGG_LDS_HASHED")) {
    h.cardSampled = std::atoi(e) == 0;
  }
  if (const char* e = std::getenv("VX355_AGG_DENSE_MIN_ROWS")) {
    h.denseMinRows = std::max<int64_t>(1, std::strtoll(e, nullptr, 10));
  }
  if (const char* e = std::getenv("VX355_AGG_RADIX_OPTIMISTIC")) {
    h.radixOptimistic = std::atoi(e) != 0;
  }
  if (const char* e = std::getenv("VX355_AGG_RADIX_OPTIMISTIC1")) {
    h.radixOptimistic1 = std::atoi(e) != 0;
  }
  if (const char* e = std::getenv("VX355_AGG_COMPACT_RECORDS")) {
    h.compactRecords = std::atoi(e) != 0;
  }
  if (const char* e = std::getenv("VX355_AGG_HASH_SLOTS")) {
    const int v = std::atoi(e);
    h.hashSlotsFixed = v <= 0 ? 0 : static_cast<int32_t>(nextPow2(static_cast<uint64_t>(std::min(kHashSlots, std::max(512, v)))));
  }
  if (const char* e = std::getenv("VX355_AGG_RADIX_SORTED")) {
    h.radixSorted = std::atoi(e) != 0;
  }
  if (const char* e = std::getenv("VX355_AGG_COARSE_LEVEL2")) {
    h.coarseLevel2 = std::atoi(e) != 0;
  }
  if (const char* e = std::getenv("VX355_AGG_FOLD_GROUPS")) {
    h.foldGroups = std::atoi(e) != 0;
  }
  if (const char* e = std::getenv("VX355_AGG_RADIX_MIN_ROWS")) {
    h.radixMinRows = std::strtoll(e, nullptr, 10);  // < 0 disables the path
  }
  if (const char* e = std::getenv("VX355_AGG_RADIX_BINS")) {
    h.radixMaxBins = std::max(2, std::min(kRadixMaxBins, std::atoi(e)));
  }
  if (const char* e = std::getenv("VX355_AGG_RADIX_TILE_ROWS")) {
    h.radixTileRows = std::strtoll(e, nullptr, 10);
  }
  if (const char* e = std::getenv("VX355_AGG_CHUNK_ROWS")) {
    h.chunkRows = std::max<int64_t>(64, std::strtoll(e, nullptr, 10) & ~63LL);
  }
}

bool needsDistinctSet(const vx355_agg_fn& f) {
  // min / max of the distinct values == min / max of the values
  return (f.flags & VX355_AGG_FN_DISTINCT) &&
      (f.kind == VX355_AGG_SUM || f.kind == VX355_AGG_COUNT || f.kind == VX355_AGG_AVG);
}

bool isStringMinMax(const vx355_agg_fn& f) {
  return (f.kind == VX355_AGG_MIN || f.kind == VX355_AGG_MAX) && isString(f.input_type);
}

vx355_agg* makeChild(vx355_agg& parent, const vx355_agg_spec& spec) {
  auto c = std::make_unique<vx355_agg>();
  c->step = spec.step;
  c->ignoreNullKeys = spec.ignore_null_keys != 0;
  configureFromEnv(*c);
  buildPlan(*c, spec);
  c->ownsCtx = false;
  return c.release();
}

// The set accumulators and their consumers of the DISTINCT aggregates of 'spec'
// (GroupingSet.cpp:117-126, DistinctAggregations::create).
void buildDistinctParts(vx355_agg& h, const vx355_agg_spec& spec) {
  const int32_t nk = spec.num_keys;
  for (int32_t i = 0; i < spec.num_aggs; ++i) {
    const vx355_agg_fn& f = spec.aggs[i];
    if (!needsDistinctSet(f) && !isStringMinMax(f)) {
      continue;
    }
    VX_CHECK_ARG(f.input_col >= 0 && f.input_col < VX355_PROJECTION_COL_BASE, "aggregate needs an input column");
    h.distinct.emplace_back();
    vx355_agg::DistinctPart& part = h.distinct.back();
    part.specIndex = i;
    part.stringMinMax = isStringMinMax(f);
    // dedup: GROUP BY keys..., x [, mask] without aggregates; null keys are values here
    std::vector<int32_t> cols(spec.key_cols, spec.key_cols + nk), types(spec.key_types, spec.key_types + nk);
    cols.push_back(f.input_col);
    types.push_back(f.input_type);
    if (f.mask_col >= 0) {
      cols.push_back(f.mask_col);
      types.push_back(VX355_BOOLEAN);
    }
    vx355_agg_spec ds{};
    ds.num_keys = static_cast<int32_t>(cols.size());
    ds.key_cols = cols.data();
    ds.key_types = types.data();
    ds.step = VX355_STEP_SINGLE;
    part.dedup = makeChild(h, ds);
    // outer: the aggregate over dedup's output columns 0..nk-1 | x | mask
    std::vector<int32_t> ocols(nk);
    for (int32_t k = 0; k < nk; ++k) {
      ocols[k] = k;
    }
    vx355_agg_fn of = f;
    of.flags = 0;
    of.input_col = nk;
    of.input_col2 = -1;
    if (part.stringMinMax) {
      of.input_type = VX355_BIGINT;  // ranks
    }
    of.mask_col = f.mask_col >= 0 ? nk + 1 : -1;
    vx355_agg_spec os{};
    os.num_keys = nk;
    os.key_cols = ocols.data();
    os.key_types = spec.key_types;
    os.num_aggs = 1;
    os.aggs = &of;
    os.step = VX355_STEP_SINGLE;
    os.ignore_null_keys = spec.ignore_null_keys;
    os.flags = spec.flags;
    part.outer = makeChild(h, os);
    part.outer->keysOptional = true;
    part.outer->unorderedOutput = false;  // rows pair up with the parent's by order
  }
}

// Ranks of the m strings of 'views' (nulls get an arbitrary one): part.rank[row], part.sortedRows[rank].
void rankStrings(vx355_agg::DistinctPart& part, const uint4* views, const uint64_t* nulls, int64_t m) {
  auto& rt = Runtime::get();
  const size_t cap = static_cast<size_t>(std::max<int64_t>(m, 1));
  uint32_t* perm = static_cast<uint32_t*>(part.perm.ensure(cap * 4 + 64));
  uint32_t* permTmp = static_cast<uint32_t*>(part.permTmp.ensure(cap * 4 + 64));
  uint64_t* keys = static_cast<uint64_t*>(part.sortKeys.ensure(cap * 8 + 64));
  uint64_t* keysTmp = static_cast<uint64_t*>(part.sortKeysTmp.ensure(cap * 8 + 64));
  int64_t* rank = static_cast<int64_t*>(part.rank.ensure(cap * 8 + 64));
  part.sortedRows = perm;
  if (m == 0) {
    return;
  }
  const int grid = static_cast<int>(ceilDiv(m, 256));
  HIP_OK(hipMemsetAsync(keys, 0, 4, rt.stream));
  VX_LAUNCH("k_str_max_len", k_str_max_len, streamGrid(m, 256), 256, 0, views, nulls, m, reinterpret_cast<uint32_t*>(keys));
  uint32_t maxLen = 0;
  copyOut(&maxLen, VX355_MEM_HOST, keys, 4);
  VX_LAUNCH("k_iota_u32", k_iota_u32, grid, 256, 0, perm, m);
  const int levels = static_cast<int>((static_cast<uint64_t>(maxLen) + 7) / 8);
  for (int level = -1, pass = 0; pass <= levels; ++pass) {
    VX_LAUNCH("k_str_sort_key", k_str_sort_key, grid, 256, 0, views, nulls, perm, m, level, keys);
    bool inTmp = false;
    sortPairsU64U32(keys, perm, keysTmp, permTmp, static_cast<size_t>(m), part.sortScratch, &inTmp, level < 0 ? 32 : 64);
    if (inTmp) {
      std::swap(keys, keysTmp);
      std::swap(perm, permTmp);
    }
    level = levels - 1 - pass;  // after the length: the last word first
  }
  part.sortedRows = perm;
  VX_LAUNCH("k_rank_from_perm", k_rank_from_perm, grid, 256, 0, perm, m, rank);
}

// noMoreInput of a min / max over strings: all of dedup's rows at once, the strings ranked,
// min / max of the ranks per group.
void pumpStringMinMax(vx355_agg& h, vx355_agg::DistinctPart& part) {
  auto& rt = Runtime::get();
  vx355_agg& dedup = *part.dedup;
  vx355_agg& outer = *part.outer;
  flushPending(dedup);
  dedup.noMoreInput = true;
  if (dedup.numOutput < 0) {
    finalize(dedup);
  }
  const int64_t total = dedup.numOutput;
  if (total > (1LL << 31) - 64) {
    VX_THROW(VX355_EUNSUPPORTED, "min / max over strings: more than 2^31 distinct (keys, string) rows");
  }
  const int32_t nc = static_cast<int32_t>(dedup.outTypes.size());
  const int32_t nk = static_cast<int32_t>(h.keys.size());
  const size_t cap = static_cast<size_t>(std::max<int64_t>(total, 1));
  const size_t words = (cap + 63) / 64;
  part.pairValues.resize(nc);
  part.pairNulls.resize(nc);
  std::vector<vx355_out_column> out(nc);
  for (int32_t c = 0; c < nc; ++c) {
    const int w = kindWidth(dedup.outTypes[c]);
    out[c].type_kind = dedup.outTypes[c];
    out[c].mem = VX355_MEM_DEVICE;
    out[c].values = part.pairValues[c].ensure((w == 0 ? words * 8 : cap * w) + 64);
    out[c].nulls = static_cast<uint64_t*>(part.pairNulls[c].ensure(words * 8 + 64));
  }
  int32_t n = 0, fin = 0;
  getOutput(dedup, out.data(), nc, static_cast<int32_t>(cap), &n, &fin);
  if (!fin) {
    VX_THROW(VX355_EINTERNAL, "dedup table listed fewer rows than it holds");
  }
  rankStrings(part, static_cast<const uint4*>(out[nk].values), out[nk].nulls, n);
  if (n > 0) {
    std::vector<vx355_column> in(nc);
    for (int32_t c = 0; c < nc; ++c) {
      in[c] = vx355_column{};
      in[c].type_kind = out[c].type_kind;
      in[c].encoding = VX355_FLAT;
      in[c].values = out[c].values;
      in[c].nulls = out[c].nulls;
      in[c].mem = VX355_MEM_DEVICE;
    }
    in[nk].type_kind = VX355_BIGINT;
    in[nk].values = part.rank.ptr();
    vx355_batch b{n, nc, in.data()};
    addInput(outer, &b);
  }
  outer.noMoreInput = true;
  rt.sync();
  // dedup stays: the views of strings longer than 12 bytes point into its arena
}

// One page of a string min / max column: outer's ranks -> views -> the caller's column.
void stringMinMaxOutput(vx355_agg& h, vx355_agg::DistinctPart& part, vx355_out_column& dst, int32_t maxRows,
                        int32_t expected) {
  auto& rt = Runtime::get();
  const int32_t nk = static_cast<int32_t>(h.keys.size());
  const size_t cap = static_cast<size_t>(std::max(maxRows, 1));
  const size_t words = (cap + 63) / 64;
  std::vector<vx355_out_column> theirs(nk + 1);
  for (int32_t k = 0; k < nk; ++k) {
    theirs[k] = vx355_out_column{h.keys[k].kind, VX355_MEM_DEVICE, nullptr, nullptr};
  }
  theirs[nk] = vx355_out_column{VX355_BIGINT, VX355_MEM_DEVICE, part.outRank.ensure(cap * 8 + 64),
                                static_cast<uint64_t*>(part.outRankNulls.ensure(words * 8 + 64))};
  int32_t n = 0, fin = 0;
  getOutput(*part.outer, theirs.data(), nk + 1, maxRows, &n, &fin);
  if (n != expected) {
    VX_THROW(VX355_EINTERNAL, "string min / max lists " + std::to_string(n) + " groups, the operator " +
                                 std::to_string(expected));
  }
  if (n == 0) {
    return;
  }
  VX_CHECK_ARG(dst.values != nullptr, "output column without values buffer");
  const bool host = dst.mem == VX355_MEM_HOST;
  uint4* views = host ? static_cast<uint4*>(part.outViews.ensure(cap * 16 + 64)) : static_cast<uint4*>(dst.values);
  uint64_t* nulls = host ? static_cast<uint64_t*>(part.outViewNulls.ensure(words * 8 + 64)) : dst.nulls;
  const int32_t sv = nk;  // dedup's columns: keys..., string [, mask]
  VX_LAUNCH("k_rank_to_view", k_rank_to_view, static_cast<int>(ceilDiv(n, 256)), 256, 0,
            static_cast<const int64_t*>(part.outRank.ptr()), static_cast<const uint64_t*>(part.outRankNulls.ptr()), n,
            part.sortedRows, static_cast<const uint4*>(part.pairValues[sv].ptr()), views, nulls);
  if (host) {
    copyOutAsync(dst.values, VX355_MEM_HOST, views, static_cast<size_t>(n) * 16);
    if (dst.nulls) {
      copyOutAsync(dst.nulls, VX355_MEM_HOST, nulls, static_cast<size_t>(ceilDiv(n, 64)) * 8);
    }
    rt.sync();
    fetchLongStrings(static_cast<char*>(dst.values), n, h.hostStrings);
  } else {
    rt.sync();
  }
}

void stringToIntermediate(vx355_agg& h, const vx355_batch* batch, const vx355_agg_fn& f, vx355_out_column& dst) {
  auto& rt = Runtime::get();
  VX_CHECK_ARG(dst.type_kind == f.input_type && dst.values != nullptr, "output column of a string min / max");
  VX_CHECK_ARG(f.input_col >= 0 && f.input_col < batch->num_cols, "aggregate input column");
  const bool host = dst.mem == VX355_MEM_HOST;
  if (!host && batch->cols[f.input_col].mem == VX355_MEM_HOST) {
    // the views of a staged host column would outlive their bytes
    VX_THROW(VX355_EUNSUPPORTED, "toIntermediate: device output of a host string column");
  }
  DeviceBatch db;
  db.load(batch, {f.input_col, f.mask_col});
  const int32_t n = db.numRows();
  if (n == 0) {
    return;
  }
  const size_t words = static_cast<size_t>(ceilDiv(n, 64));
  DevBuf dViews, dNulls;
  uint4* views = host ? static_cast<uint4*>(dViews.ensure(static_cast<size_t>(n) * 16 + 64)) : static_cast<uint4*>(dst.values);
  uint64_t* nulls = host ? static_cast<uint64_t*>(dNulls.ensure(words * 8 + 64)) : dst.nulls;
  const ColView in = db.col(f.input_col);
  const ColView mask = f.mask_col >= 0 ? db.col(f.mask_col) : ColView{};
  VX_LAUNCH("k_string_pass", k_string_pass, static_cast<int>(ceilDiv(n, 256)), 256, 0, in, mask, f.mask_col >= 0 ? 1 : 0, n,
            views, nulls);
  if (host) {
    copyOutAsync(dst.values, VX355_MEM_HOST, views, static_cast<size_t>(n) * 16);
    if (dst.nulls) {
      copyOutAsync(dst.nulls, VX355_MEM_HOST, nulls, words * 8);
    }
    rt.sync();
    fetchLongStrings(static_cast<char*>(dst.values), n, h.hostStrings);
  } else {
    rt.sync();
  }
}

void feedInput(vx355_agg& h, const vx355_batch* batch) {
  if (!tryCoalesce(h, batch)) {
    flushPending(h);
    addInput(h, batch);
  }
}

// noMoreInput of one DISTINCT aggregate: dedup's rows -> outer (TypedDistinctAggregations::
// extractValues, DistinctAggregations.cpp:240-285, with the per-group loop turned into one more
// aggregation).
void pumpDistinct(vx355_agg& h, vx355_agg::DistinctPart& part) {
  vx355_agg& dedup = *part.dedup;
  vx355_agg& outer = *part.outer;
  flushPending(dedup);
  dedup.noMoreInput = true;
  const int32_t nc = static_cast<int32_t>(dedup.outTypes.size());
  const int32_t chunk = 1 << 22;
  const size_t words = static_cast<size_t>(chunk / 64);
  std::vector<DevBuf> values(nc), nulls(nc);
  std::vector<vx355_out_column> out(nc);
  std::vector<vx355_column> in(nc);
  bool allocated = false;
  for (;;) {
    if (!allocated) {
      for (int32_t c = 0; c < nc; ++c) {
        const int w = kindWidth(dedup.outTypes[c]);
        out[c].type_kind = dedup.outTypes[c];
        out[c].mem = VX355_MEM_DEVICE;
        out[c].values = values[c].ensure(w == 0 ? words * 8 : static_cast<size_t>(chunk) * w);
        out[c].nulls = static_cast<uint64_t*>(nulls[c].ensure(words * 8));
      }
      allocated = true;
    }
    int32_t n = 0, fin = 0;
    getOutput(dedup, out.data(), nc, chunk, &n, &fin);
    if (n > 0) {
      for (int32_t c = 0; c < nc; ++c) {
        in[c] = vx355_column{};
        in[c].type_kind = out[c].type_kind;
        in[c].encoding = VX355_FLAT;
        in[c].values = out[c].values;
        in[c].nulls = out[c].nulls;
        in[c].mem = VX355_MEM_DEVICE;
      }
      vx355_batch b{n, nc, in.data()};
      addInput(outer, &b);
    }
    if (fin) {
      break;
    }
  }
  outer.noMoreInput = true;
  delete part.dedup;  // the outer operator has copied what it keeps (long strings included)
  part.dedup = nullptr;
}

}  // namespace
}  // namespace vx

namespace vx {
// Output columns that hold the SUM half of an avg's intermediate (sum, count) pair - the PARTIAL /
// INTERMEDIATE layout ships avg as two flat columns; on the wire the pair is Velox's
// ROW(DOUBLE sum, BIGINT count) (functions/lib/aggregates/AverageAggregateBase.h:66-260).
std::vector<int32_t> aggPartialAvgColumns(const vx355_agg* h) {
  std::vector<int32_t> out;
  if (finalOutput(h->step)) {
    return out;
  }
  int32_t col = static_cast<int32_t>(h->keys.size());
  const auto& fns = h->distinct.empty() ? std::vector<vx355_agg_fn>() : h->specAggs;
  if (!h->distinct.empty()) {
    for (const auto& f : fns) {
      if (f.kind == VX355_AGG_AVG) {
        out.push_back(col);
        ++col;
      }
      ++col;
    }
    return out;
  }
  for (const auto& la : h->aggs) {
    if (la.fn.kind == VX355_AGG_AVG) {
      out.push_back(col);
      ++col;
    }
    ++col;
  }
  return out;
}
}  // namespace vx

extern "C" {

int vx355_agg_create(const vx355_agg_spec* spec, vx355_agg** out) {
  VX_API_BEGIN
  Runtime::get().requireInit();
  VX_CHECK_ARG(spec && out, "NULL argument");
  VX_CHECK_ARG(spec->num_keys >= 0 && spec->num_aggs >= 0, "negative counts");
  VX_CHECK_ARG(spec->step >= VX355_STEP_PARTIAL && spec->step <= VX355_STEP_SINGLE, "bad step");
  auto h = std::make_unique<vx355_agg>();
  h->step = spec->step;
  h->ignoreNullKeys = spec->ignore_null_keys != 0;
  h->unorderedOutput = (spec->flags & VX355_AGG_UNORDERED_OUTPUT) != 0;
  configureFromEnv(*h);
  // DISTINCT aggregates get their own tables; this operator keeps the others.
  std::vector<vx355_agg_fn> plain;
  bool anyDistinct = false, anyParts = false;
  for (int32_t i = 0; i < spec->num_aggs; ++i) {
    vx355_agg_fn f = spec->aggs[i];
    const bool ds = needsDistinctSet(f);
    const bool d = ds || isStringMinMax(f);
    if ((f.flags & VX355_AGG_FN_DISTINCT) && f.kind == VX355_AGG_COUNT_STAR) {
      VX_THROW(VX355_EINVAL, "count(*) has no input to be DISTINCT over");
    }
    anyDistinct = anyDistinct || ds;
    anyParts = anyParts || d;
    h->specIsDistinct.push_back(d ? 1 : 0);
    if (!d) {
      f.flags &= ~VX355_AGG_FN_DISTINCT;
      plain.push_back(f);
    }
  }
  if (anyDistinct && spec->step != VX355_STEP_SINGLE) {
    // GroupingSet.cpp:117-121
    VX_THROW(VX355_EUSER, "Partial aggregations over distinct inputs are not supported");
  }
  if (anyParts) {
    vx355_agg_spec mainSpec = *spec;
    mainSpec.num_aggs = static_cast<int32_t>(plain.size());
    mainSpec.aggs = plain.data();
    buildPlan(*h, mainSpec);
    h->unorderedOutput = false;  // the parts pair up with this operator's rows by order
    h->specKeyCols.assign(spec->key_cols, spec->key_cols + spec->num_keys);
    h->specKeyTypes.assign(spec->key_types, spec->key_types + spec->num_keys);
    h->specAggs.assign(spec->aggs, spec->aggs + spec->num_aggs);
    h->specFlags = spec->flags;
    buildDistinctParts(*h, *spec);
    h->fullOutTypes.assign(h->outTypes.begin(), h->outTypes.begin() + spec->num_keys);
    size_t nextPlain = static_cast<size_t>(spec->num_keys), nextPart = 0;
    for (int32_t i = 0; i < spec->num_aggs; ++i) {
      h->specColBegin.push_back(static_cast<int32_t>(h->fullOutTypes.size()));
      if (h->specIsDistinct[i]) {
        const auto& part = h->distinct[nextPart++];
        h->fullOutTypes.push_back(part.stringMinMax ? spec->aggs[i].input_type : part.outer->outTypes.back());
      } else {
        // avg has two columns (sum, count) in the partial / intermediate layout
        const int width = (spec->aggs[i].kind == VX355_AGG_AVG && !finalOutput(spec->step)) ? 2 : 1;
        for (int w = 0; w < width; ++w) {
          h->fullOutTypes.push_back(h->outTypes[nextPlain++]);
        }
      }
    }
    h->specColBegin.push_back(static_cast<int32_t>(h->fullOutTypes.size()));
  } else {
    buildPlan(*h, *spec);
  }
  h->ctx = Runtime::createContext();  // the operator's own stream + mailbox
  for (auto& d : h->distinct) {
    d.dedup->ctx = h->ctx;
    d.outer->ctx = h->ctx;
  }
  *out = h.release();
  VX_API_END
}

int vx355_agg_set_fused_input(vx355_agg* h, const vx355_filter_term* terms, int32_t n_terms,
                              const vx355_projection* projections, int32_t n_projections) {
  VX_ASYNC_DRAIN(h)
  VX_API_BEGIN_CTX(VX_CTX_OF(h))
  VX_CHECK_ARG(h, "NULL argument");
  VX_CHECK_ARG(h->inputRows == 0 && !h->tableReady, "set_fused_input after the first add_input");
  VX_CHECK_ARG(n_terms >= 0 && n_terms <= kMaxTerms && (n_terms == 0 || terms), "0..4 filter terms");
  VX_CHECK_ARG(n_projections >= 0 && n_projections <= kMaxProjections &&
                   (n_projections == 0 || projections),
               "0..4 projections");
  if (!rawInput(h->step)) {
    VX_THROW(VX355_EUNSUPPORTED, "fused FilterProject needs raw input (partial or single step)");
  }
  if (!h->distinct.empty()) {
    VX_THROW(VX355_EUNSUPPORTED, "fused FilterProject under DISTINCT aggregates");
  }
  h->fusedTerms.assign(terms, terms + n_terms);
  h->fusedProj.assign(projections, projections + n_projections);
  for (const auto& t : h->fusedTerms) {
    VX_CHECK_ARG(t.col >= 0, "filter term without a column");
    h->usedCols.push_back(t.col);
  }
  for (const auto& p : h->fusedProj) {
    VX_CHECK_ARG(p.num_factors >= 1 && p.num_factors <= kMaxFactors, "1..4 factors per projection");
    for (int f = 0; f < p.num_factors; ++f) {
      h->usedCols.push_back(p.factors[f].col);
    }
  }
  VX_API_END
}

int vx355_agg_add_input(vx355_agg* h, const vx355_batch* batch) {
  VX_ASYNC_DRAIN(h)
  VX_API_BEGIN_CTX(VX_CTX_OF(h))
  Runtime::get().requireInit();
  VX_CHECK_ARG(h && batch, "NULL argument");
  VX_CHECK_ARG(!h->noMoreInput, "addInput after noMoreInput");
  VX_CHECK_ARG(!h->flushing, "addInput while a partial flush is being drained");
  feedInput(*h, batch);
  for (auto& d : h->distinct) {
    feedInput(*d.dedup, batch);
  }
  publishStats(*h);
  VX_API_END
}

// (the entry point minus its drain: what the queue's worker runs for vx355_agg_no_more_input_async)
static int aggNoMoreInputNow(vx355_agg* h) {
  VX_API_BEGIN_CTX(VX_CTX_OF(h))
  VX_CHECK_ARG(h, "NULL argument");
  Runtime::get().requireInit();
  flushPending(*h);
  if (!h->noMoreInput) {
    for (auto& d : h->distinct) {
      if (d.stringMinMax) {
        pumpStringMinMax(*h, d);
      } else {
        pumpDistinct(*h, d);
      }
    }
  }
  h->noMoreInput = true;
  VX_API_END
}

int vx355_agg_no_more_input(vx355_agg* h) {
  VX_ASYNC_DRAIN(h)
  return aggNoMoreInputNow(h);
}

int vx355_agg_output_types(const vx355_agg* h, int32_t* types, int32_t cap, int32_t* n) {
  VX_API_BEGIN
  VX_CHECK_ARG(h && n, "NULL argument");
  const std::vector<int32_t>& seen = h->distinct.empty() ? h->outTypes : h->fullOutTypes;
  *n = static_cast<int32_t>(seen.size());
  if (types) {
    for (int32_t i = 0; i < *n && i < cap; ++i) {
      types[i] = seen[i];
    }
  }
  VX_API_END
}

static int aggGetOutputNow(vx355_agg* h, vx355_out_column* cols, int32_t num_cols, int32_t max_rows,
                           int32_t* n_out, int32_t* finished) {
  VX_API_BEGIN_CTX(VX_CTX_OF(h))
  Runtime::get().requireInit();
  VX_CHECK_ARG(h, "NULL argument");
  if (h->distinct.empty()) {
    getOutput(*h, cols, num_cols, max_rows, n_out, finished);
  } else {
    VX_CHECK_ARG(cols && n_out && finished, "NULL argument");
    VX_CHECK_ARG(num_cols == static_cast<int32_t>(h->fullOutTypes.size()), "wrong number of output columns");
    const int32_t nk = static_cast<int32_t>(h->keys.size());
    std::vector<vx355_out_column> mine(cols, cols + nk);
    for (size_t i = 0; i < h->specIsDistinct.size(); ++i) {
      if (!h->specIsDistinct[i]) {
        mine.insert(mine.end(), cols + h->specColBegin[i], cols + h->specColBegin[i + 1]);
      }
    }
    h->hostStrings.clear();
    const bool wasFlushing = h->flushing;
    const bool keepStrings = h->generic && h->hasStringKeys;  // getOutput clears the list itself then
    std::vector<std::vector<char>> keyStrings;
    getOutput(*h, mine.data(), static_cast<int32_t>(mine.size()), max_rows, n_out, finished);
    if (keepStrings) {
      keyStrings.swap(h->hostStrings);
    }
    for (auto& d : h->distinct) {
      if (d.stringMinMax) {
        const int32_t at = h->specColBegin[d.specIndex];
        VX_CHECK_ARG(cols[at].type_kind == h->fullOutTypes[at], "output column type mismatch");
        stringMinMaxOutput(*h, d, cols[at], max_rows, *n_out);
        continue;
      }
      std::vector<vx355_out_column> theirs(nk + 1);
      for (int32_t k = 0; k < nk; ++k) {
        theirs[k] = vx355_out_column{h->keys[k].kind, VX355_MEM_DEVICE, nullptr, nullptr};
      }
      theirs[nk] = cols[h->specColBegin[d.specIndex]];
      int32_t n2 = 0, fin2 = 0;
      getOutput(*d.outer, theirs.data(), nk + 1, max_rows, &n2, &fin2);
      if (n2 != *n_out) {
        VX_THROW(VX355_EINTERNAL, "DISTINCT aggregate lists " + std::to_string(n2) + " groups, the operator " +
                                     std::to_string(*n_out));
      }
    }
    for (auto& block : keyStrings) {
      h->hostStrings.push_back(std::move(block));
    }
    if (wasFlushing && !h->flushing) {
      // the flush has been drained (getOutput reset this operator's table): fresh parts. The
      // old ones stay until the next flush: device output columns point into their arenas.
      h->dropParts(h->retired);
      h->retired.swap(h->distinct);
      vx355_agg_spec spec{};
      spec.num_keys = static_cast<int32_t>(h->specKeyCols.size());
      spec.key_cols = h->specKeyCols.data();
      spec.key_types = h->specKeyTypes.data();
      spec.num_aggs = static_cast<int32_t>(h->specAggs.size());
      spec.aggs = h->specAggs.data();
      spec.step = h->step;
      spec.ignore_null_keys = h->ignoreNullKeys ? 1 : 0;
      spec.flags = h->specFlags;
      buildDistinctParts(*h, spec);
      for (auto& d : h->distinct) {
        d.dedup->ctx = h->ctx;
        d.outer->ctx = h->ctx;
      }
    }
  }
  VX_API_END
}

int vx355_agg_get_output(vx355_agg* h, vx355_out_column* cols, int32_t num_cols, int32_t max_rows,
                         int32_t* n_out, int32_t* finished) {
  VX_ASYNC_DRAIN(h)
  return aggGetOutputNow(h, cols, num_cols, max_rows, n_out, finished);
}

// Queued forms (ABI 7): no_more_input and get_output as tasks of the handle's worker, behind the batches
// already queued. The Driver thread neither waits for the last batches nor for the listing of the groups and
// the copies into the result vectors; 'done' fires on the worker thread when the page is there.
int vx355_agg_no_more_input_async(vx355_agg* h, int64_t* ticket_out) {
  try {
    if (!h) {
      vx::setLastError("NULL argument");
      return VX355_EINVAL;
    }
    if (!h->aq) {
      h->aq = vx::asyncCreate();
    }
    if (const int failed = vx::asyncFailed(h->aq)) {
      return failed;
    }
    vx::asyncSealIngest(h->aq);
    const int64_t ticket = vx::asyncSubmit(h->aq, [h](std::string* text) {
      const int status = aggNoMoreInputNow(h);
      if (status != VX355_OK) {
        *text = vx355_last_error();
      }
      return status;
    });
    if (ticket_out) {
      *ticket_out = ticket;
    }
    return VX355_OK;
  } catch (const std::exception& e) {
    vx::setLastError(e.what());
    return VX355_EINTERNAL;
  }
}

int vx355_agg_get_output_async(vx355_agg* h, const vx355_out_column* cols, int32_t num_cols, int32_t max_rows,
                               vx355_output_done_fn done, void* done_arg, int64_t* ticket_out) {
  try {
    if (!h || (num_cols > 0 && !cols)) {
      vx::setLastError("NULL argument");
      return VX355_EINVAL;
    }
    if (!h->aq) {
      h->aq = vx::asyncCreate();
    }
    if (const int failed = vx::asyncFailed(h->aq)) {
      return failed;
    }
    vx::asyncSealIngest(h->aq);
    auto page = std::make_shared<vx355_agg::QueuedPage>();
    page->cols.assign(cols, cols + num_cols);
    std::function<void(int)> fire;
    if (done) {
      fire = [page, done, done_arg](int queueStatus) {
        done(done_arg, page->status != VX355_OK ? page->status : queueStatus, page->numRows, page->finished);
      };
    }
    {
      // (registered before the task can run: the ticket is only known after the submit)
      std::lock_guard<std::mutex> lock(h->pagesMutex);
      const int64_t ticket = vx::asyncSubmit(
          h->aq,
          [h, page, max_rows](std::string* text) {
            page->status = aggGetOutputNow(h, page->cols.data(), static_cast<int32_t>(page->cols.size()), max_rows,
                                           &page->numRows, &page->finished);
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
    }
    return VX355_OK;
  } catch (const std::exception& e) {
    vx::setLastError(e.what());
    return VX355_EINTERNAL;
  }
}

int vx355_agg_output_result(vx355_agg* h, int64_t ticket, int32_t* n_out, int32_t* finished) {
  if (!h || !n_out || !finished) {
    vx::setLastError("NULL argument");
    return VX355_EINVAL;
  }
  std::shared_ptr<vx355_agg::QueuedPage> page;
  {
    std::lock_guard<std::mutex> lock(h->pagesMutex);
    auto it = h->pages.find(ticket);
    if (it == h->pages.end()) {
      vx::setLastError("no queued get_output with this ticket (results are handed out once)");
      return VX355_EINVAL;
    }
    // The queue's position FIRST, the page's flag after it: read the other way round, a worker that finishes
    // the page between the two reads would make a completed page look like one skipped behind a failure.
    int64_t submitted = 0, completed = 0;
    vx::asyncPoll(h->aq, &submitted, &completed);
    if (!it->second->complete.load(std::memory_order_acquire)) {
      // a page skipped behind a failed batch never runs: the queue's failure is the answer then
      if (completed < ticket || vx::asyncFailed(h->aq) == VX355_OK) {
        vx::setLastError("the queued get_output has not completed (vx355_agg_poll: completed < ticket)");
        return VX355_EINVAL;
      }
      h->pages.erase(it);
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

int vx355_agg_flush(vx355_agg* h) {
  VX_ASYNC_DRAIN(h)
  VX_API_BEGIN_CTX(VX_CTX_OF(h))
  VX_CHECK_ARG(h, "NULL argument");
  VX_CHECK_ARG(!h->noMoreInput, "flush after noMoreInput");
  if (finalOutput(h->step) || h->keys.empty()) {
    // HashAggregation.cpp:218-224: only partial output is flushed, never a global aggregation
    VX_THROW(VX355_EINVAL, "flush applies to partial / intermediate steps with grouping keys");
  }
  flushPending(*h);
  // min / max over strings: their tables are closed like at noMoreInput and rebuilt once the
  // flushed groups are drained (DISTINCT parts only exist in the SINGLE step)
  for (auto& d : h->distinct) {
    pumpStringMinMax(*h, d);
  }
  h->flushing = true;
  h->numOutput = -1;
  h->outputCursor = 0;
  VX_API_END
}

int vx355_agg_to_intermediate(vx355_agg* h, const vx355_batch* batch, vx355_out_column* cols, int32_t num_cols) {
  VX_ASYNC_DRAIN(h)
  VX_API_BEGIN_CTX(VX_CTX_OF(h))
  VX_CHECK_ARG(h, "NULL argument");
  if (h->distinct.empty()) {
    toIntermediate(*h, batch, cols, num_cols);
  } else {
    // min / max over strings pass their input through; the other aggregates go the usual way
    VX_CHECK_ARG(batch && cols, "NULL argument");
    std::vector<vx355_out_column> plain;
    std::vector<std::pair<size_t, int32_t>> strings;  // aggregate, output column
    int32_t c = 0;
    for (size_t i = 0; i < h->specAggs.size(); ++i) {
      const int32_t width = h->specAggs[i].kind == VX355_AGG_AVG ? 2 : 1;  // the PARTIAL layout
      VX_CHECK_ARG(c + width <= num_cols, "wrong number of aggregate output columns");
      if (h->specIsDistinct[i]) {
        if (!isStringMinMax(h->specAggs[i])) {
          VX_THROW(VX355_EUNSUPPORTED, "toIntermediate with DISTINCT aggregates");
        }
        strings.emplace_back(i, c);
      } else {
        plain.insert(plain.end(), cols + c, cols + c + width);
      }
      c += width;
    }
    VX_CHECK_ARG(c == num_cols, "wrong number of aggregate output columns");
    if (!plain.empty()) {
      toIntermediate(*h, batch, plain.data(), static_cast<int32_t>(plain.size()));
    } else if (!rawInput(h->step)) {
      VX_THROW(VX355_EINVAL, "toIntermediate applies to raw input (partial / single steps); intermediate input passes through");
    }
    h->hostStrings.clear();
    for (const auto& job : strings) {
      stringToIntermediate(*h, batch, h->specAggs[job.first], cols[job.second]);
    }
  }
  VX_API_END
}

int vx355_agg_get_stats(const vx355_agg* h, vx355_agg_stats* out) {
  if (h != nullptr && h->aq != nullptr) {
    vx::asyncQuiesce(h->aq);  // (inspection: readable on a handle whose queue has failed)
  }
  VX_API_BEGIN
  VX_CHECK_ARG(h && out, "NULL argument");
  out->num_groups = h->keys.empty() ? 1 : h->numGroups;
  out->capacity = static_cast<int64_t>(h->capacity);
  out->num_rehashes = h->numRehashes;
  out->hash_mode = h->mode;
  out->reserved = static_cast<int32_t>(std::min<int64_t>(h->jitLaunches, INT32_MAX));
  out->radix_launches = h->radixLaunches;
  out->input_rows = h->inputRows + h->coalescer.pendingRows();
  out->deferred_rows = h->deferredRows;
  out->table_bytes = tableBytesOf(*h);
  out->num_flushes = h->numFlushes;
  out->compact_record_launches = h->compactLaunches;
  VX_API_END
}

int vx355_agg_get_gpu_stats(const vx355_agg* h, vx355_gpu_stats* out) { return vx::gpuStatsOf(h ? h->ctx : nullptr, out); }

int vx355_agg_bytes_in_use(const vx355_agg* h, int64_t* bytes) {
  if (!h || !bytes) {
    vx::setLastError("NULL argument");
    return VX355_EINVAL;
  }
  *bytes = h->publishedUsedBytes.load(std::memory_order_relaxed);
  return VX355_OK;
}

int vx355_agg_table_bytes(const vx355_agg* h, int64_t* table_bytes, int64_t* num_groups) {
  if (!h) {
    vx::setLastError("NULL argument");
    return VX355_EINVAL;
  }
  if (table_bytes) {
    *table_bytes = h->publishedTableBytes.load(std::memory_order_relaxed);
  }
  if (num_groups) {
    *num_groups = h->publishedGroups.load(std::memory_order_relaxed);
  }
  return VX355_OK;
}

void* vx355_agg_stream(vx355_agg* h) { return h ? static_cast<void*>(h->ctx->stream) : nullptr; }

void vx355_agg_destroy(vx355_agg* h) {
  if (!h) {
    return;
  }
  vx::asyncDestroy(h->aq);  // waits for the batches in flight
  h->aq = nullptr;
  Runtime* ctx = h->ctx;
  try {
    vx::ContextScope scope(ctx);  // the handle's buffers are released under its own context
    delete h;
  } catch (...) {
  }
  Runtime::destroyContext(ctx);
}

// ---- asynchronous boundary (async.hip) ----
int vx355_agg_add_input_async(vx355_agg* h, const vx355_batch* batch, int64_t* ticket_out) {
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
    const int64_t ticket = vx::asyncSubmitBatch(h->aq, h->ctx->ds, batch, h->usedCols, [h](const vx355_batch* b) -> int {
      // the synchronous entry point minus its drain (this IS the queue's worker)
      VX_API_BEGIN_CTX(VX_CTX_OF(h))
      Runtime::get().requireInit();
      VX_CHECK_ARG(!h->noMoreInput, "addInput after noMoreInput");
      VX_CHECK_ARG(!h->flushing, "addInput while a partial flush is being drained");
      feedInput(*h, b);
      for (auto& d : h->distinct) {
        feedInput(*d.dedup, b);
      }
      publishStats(*h);
      VX_API_END
    });
    if (ticket_out) {
      *ticket_out = ticket;
    }
    return VX355_OK;
  } catch (const std::exception& e) {
    vx::setLastError(e.what());
    return VX355_EINTERNAL;
  }
}

int vx355_agg_poll(vx355_agg* h, int64_t* submitted, int64_t* completed) {
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

int vx355_agg_wait(vx355_agg* h) {
  if (!h) {
    vx::setLastError("NULL argument");
    return VX355_EINVAL;
  }
  return h->aq ? vx::asyncWait(h->aq) : VX355_OK;
}

}  // extern "C"
