// The library's own LSD radix sort (8 bits per pass, stable) behind sortPairsU64U32 / sortKeysU64:
// first-seen order of fewer than 32 M groups, the hit re-ordering of the range-partitioned probe,
// counting joins, value dictionaries, string ranks. (Until round 5 these went through the vendor's
// device-wide sort.)
// None of them is on a per-row path; they run once per batch or once per operator at output time
// on thousands to a few million entries, so the design favours few, simple launches:
//   n <= 2048      one workgroup ranks every entry by counting (k_rs_small);
//   otherwise      per pass: k_rs_hist (every wave counts the digits of ITS 2048 consecutive entries),
//                  one exclusive scan over [digit][chunk] (one workgroup up to 16 K cells, else the
//                  three-launch scanU32ToU64), k_rs_scatter (the same waves place their entries: rank
//                  among the 64 lanes by eight ballots, running offsets per digit in LDS).
// Passes cover the bit range the caller names: callers that know their key width (first rows < input
// rows) or that only order by a field of the word (the probe row of a {probe row, build row} hit) say so.
#include <cstring>

#include "common.h"
#include "device_utils.h"

namespace vx {
namespace {

// Orders a wave's own LDS traffic across a point of the program (the wave's lanes run in lockstep;
// this keeps the compiler from moving LDS accesses across it and drains the ones in flight).
__device__ inline void waveSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);  // vmcnt(63) expcnt(7) lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kRsChunk = 2048;      // entries per wave
constexpr int kRsWaves = 4;         // waves per workgroup
constexpr int kRsSmall = 2048;      // entries the one-workgroup kernel takes
constexpr int64_t kRsSingleScan = 16 << 10;   // cells one workgroup scans (n <= 128 K entries); more: scanU32ToU64

__global__ __launch_bounds__(256) void k_rs_hist(const uint64_t* keys, int64_t n, int shift, uint32_t digitMask, uint32_t* hist, int64_t numChunks) {
  __shared__ uint32_t h[kRsWaves][256];
  const int w = threadIdx.x >> 6;
  const int64_t chunk = static_cast<int64_t>(blockIdx.x) * kRsWaves + w;
  for (int d = lane(); d < 256; d += 64) {
    h[w][d] = 0;
  }
  blockSync();
  if (chunk < numChunks) {
    const int64_t base = chunk * kRsChunk;
#pragma unroll 4
    for (int it = 0; it < kRsChunk / 64; ++it) {
      const int64_t i = base + it * 64 + lane();
      if (i < n) {
        atomicAdd(&h[w][(keys[i] >> shift) & digitMask], 1u);
      }
    }
  }
  blockSync();
  if (chunk < numChunks) {
    for (int d = lane(); d < 256; d += 64) {
      hist[static_cast<int64_t>(d) * numChunks + chunk] = h[w][d];
    }
  }
}

// Exclusive scan of 'n' u32 cells into u64 offsets (n + 1 entries) by one workgroup.
__global__ __launch_bounds__(1024) void k_rs_scan(const uint32_t* in, int64_t n, uint64_t* out) {
  __shared__ uint64_t partial[1024];
  const int t = threadIdx.x;
  const int64_t per = (n + 1023) / 1024;
  const int64_t begin = t * per;
  const int64_t end = begin + per < n ? begin + per : n;
  uint64_t sum = 0;
  for (int64_t i = begin; i < end; ++i) {
    sum += in[i];
  }
  partial[t] = sum;
  blockSync();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint64_t v = t >= off ? partial[t - off] : 0;
    blockSync();
    partial[t] += v;
    blockSync();
  }
  uint64_t run = t == 0 ? 0 : partial[t - 1];
  for (int64_t i = begin; i < end; ++i) {
    const uint32_t v = in[i];
    out[i] = run;
    run += v;
  }
  if (t == 1023) {
    out[n] = partial[1023];
  }
}

template <bool PAIRS>
__global__ __launch_bounds__(256) void k_rs_scatter(const uint64_t* keys, const uint32_t* vals, uint64_t* keysOut,
                                                    uint32_t* valsOut, int64_t n, int shift, uint32_t digitMask,
                                                    const uint64_t* offsets, int64_t numChunks) {
  __shared__ unsigned long long next[kRsWaves][256];  // where the wave's next entry with digit d goes
  const int w = threadIdx.x >> 6;
  const int64_t chunk = static_cast<int64_t>(blockIdx.x) * kRsWaves + w;
  if (chunk >= numChunks) {
    return;  // (no workgroup barrier below: waves are independent)
  }
  for (int d = lane(); d < 256; d += 64) {
    next[w][d] = offsets[static_cast<int64_t>(d) * numChunks + chunk];
  }
  waveSync();
  const int64_t base = chunk * kRsChunk;
  const uint64_t below = (1ULL << lane()) - 1;
  for (int it = 0; it < kRsChunk / 64; ++it) {
    const int64_t i = base + it * 64 + lane();
    const bool valid = i < n;
    const uint64_t key = valid ? keys[i] : 0;
    uint32_t val = 0;
    if (PAIRS && valid) {
      val = vals[i];
    }
    const uint32_t d = static_cast<uint32_t>(key >> shift) & digitMask;
    // lanes holding the same digit (match-any by eight ballots)
    uint64_t same = ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t m = ballot(valid && bit);
      same &= bit ? m : ~m;
    }
    const uint32_t rank = static_cast<uint32_t>(popc64(same & below));
    unsigned long long pos = 0;
    if (valid) {
      pos = next[w][d] + rank;
    }
    waveSync();  // every lane has read the offsets of this step before the leaders advance them
    if (valid) {
      keysOut[pos] = key;
      if (PAIRS) {
        valsOut[pos] = val;
      }
      if (rank == 0) {
        next[w][d] += static_cast<unsigned long long>(popc64(same));
      }
    }
    waveSync();
  }
}

// n <= kRsSmall: every entry's position = entries with a smaller key (on the low 'mask' bits) + equal
// keys in front of it. One workgroup; the keys sit in LDS and are read as broadcasts.
template <bool PAIRS>
__global__ __launch_bounds__(1024) void k_rs_small(const uint64_t* keys, const uint32_t* vals, uint64_t* keysOut,
                                                   uint32_t* valsOut, uint32_t n, uint64_t mask) {
  __shared__ uint64_t k[kRsSmall];
  for (uint32_t i = threadIdx.x; i < n; i += 1024) {
    k[i] = keys[i] & mask;
  }
  blockSync();
  for (uint32_t i = threadIdx.x; i < n; i += 1024) {
    const uint64_t mine = k[i];
    uint32_t pos = 0;
    for (uint32_t j = 0; j < n; ++j) {
      const uint64_t o = k[j];
      pos += (o < mine || (o == mine && j < i)) ? 1u : 0u;
    }
    keysOut[pos] = keys[i];
    if (PAIRS) {
      valsOut[pos] = vals[i];
    }
  }
}

// One pass over bits [shift, min(shift + 8, endBit)): (srcKeys, srcVals) -> (dstKeys, dstVals). The last
// pass of a sort masks its digit at endBit, so that bits above it never order anything - the same
// contract as the one-workgroup path (k_rs_small), whatever the callers keep up there.
template <bool PAIRS>
void radixPass(const uint64_t* srcKeys, const uint32_t* srcVals, uint64_t* dstKeys, uint32_t* dstVals, int64_t n, int shift,
               int endBit, uint32_t* hist, uint64_t* offsets, DevBuf& scanScratch) {
  const uint32_t digitMask = endBit - shift >= 8 ? 255u : ((1u << (endBit - shift)) - 1u);
  const int64_t numChunks = ceilDiv(n, kRsChunk);
  const int grid = static_cast<int>(ceilDiv(numChunks, kRsWaves));
  const int64_t cells = numChunks * 256;
  VX_LAUNCH("k_rs_hist", k_rs_hist, grid, 256, 0, srcKeys, n, shift, digitMask, hist, numChunks);
  if (cells <= kRsSingleScan) {
    VX_LAUNCH("k_rs_scan", k_rs_scan, 1, 1024, 0, hist, cells, offsets);
  } else {
    scanU32ToU64(hist, cells, offsets, scanScratch);
  }
  VX_LAUNCH("k_rs_scatter", (k_rs_scatter<PAIRS>), grid, 256, 0, srcKeys, srcVals, dstKeys, dstVals, n, shift, digitMask,
            offsets, numChunks);
}

struct SortScratch {
  uint32_t* hist;
  uint64_t* offsets;
  uint64_t* spare;  // n keys (sortKeysU64 only)
};

SortScratch carve(DevBuf& tmp, int64_t n, bool spareKeys) {
  const int64_t cells = ceilDiv(n, kRsChunk) * 256;
  const size_t histBytes = (static_cast<size_t>(cells) * 4 + 255) & ~static_cast<size_t>(255);
  const size_t offBytes = (static_cast<size_t>(cells + 1) * 8 + 255) & ~static_cast<size_t>(255);
  const size_t spareBytes = spareKeys ? static_cast<size_t>(n) * 8 + 256 : 0;
  char* base = static_cast<char*>(tmp.ensure(histBytes + offBytes + spareBytes + 64));
  return SortScratch{reinterpret_cast<uint32_t*>(base), reinterpret_cast<uint64_t*>(base + histBytes),
                     reinterpret_cast<uint64_t*>(base + histBytes + offBytes)};
}

}  // namespace

// Stable ascending sort of (key, value) pairs on key bits [0, endBit). The result lands in
// (keysTmp, valsTmp) when *resultInTmp, else in (keys, vals); both pairs of buffers are overwritten.
void sortPairsU64U32(uint64_t* keys, uint32_t* vals, uint64_t* keysTmp, uint32_t* valsTmp,
                     size_t n, DevBuf& tmp, bool* resultInTmp, int endBit) {
  *resultInTmp = false;
  if (n <= 1) {
    return;
  }
  endBit = std::max(1, std::min(64, endBit));
  if (n <= static_cast<size_t>(kRsSmall)) {
    const uint64_t mask = endBit >= 64 ? ~0ULL : ((1ULL << endBit) - 1);
    VX_LAUNCH("k_rs_small", (k_rs_small<true>), 1, 1024, 0, keys, vals, keysTmp, valsTmp, static_cast<uint32_t>(n), mask);
    *resultInTmp = true;
    return;
  }
  const SortScratch s = carve(tmp, static_cast<int64_t>(n), false);
  DevBuf scanScratch;
  const int passes = (endBit + 7) / 8;
  uint64_t* k[2] = {keys, keysTmp};
  uint32_t* v[2] = {vals, valsTmp};
  for (int p = 0; p < passes; ++p) {
    radixPass<true>(k[p & 1], v[p & 1], k[(p + 1) & 1], v[(p + 1) & 1], static_cast<int64_t>(n), p * 8, endBit, s.hist,
                    s.offsets, scanScratch);
  }
  *resultInTmp = (passes & 1) != 0;
  Runtime::get().sync();  // (scanScratch goes out of scope: the passes must have finished with it)
}

// Ascending sort of n u64 keys on bits [beginBit, endBit) (stable on the others): in -> out (both n
// entries, distinct buffers; 'in' is left untouched).
void sortKeysU64(const uint64_t* in, uint64_t* out, size_t n, DevBuf& tmp, int beginBit, int endBit) {
  if (n == 0) {
    return;
  }
  beginBit = std::max(0, std::min(63, beginBit));
  endBit = std::max(beginBit + 1, std::min(64, endBit));
  if (n <= static_cast<size_t>(kRsSmall)) {
    const uint64_t high = endBit >= 64 ? ~0ULL : ((1ULL << endBit) - 1);
    const uint64_t mask = high & ~((1ULL << beginBit) - 1);
    VX_LAUNCH("k_rs_small", (k_rs_small<false>), 1, 1024, 0, in, static_cast<const uint32_t*>(nullptr), out,
              static_cast<uint32_t*>(nullptr), static_cast<uint32_t>(n), mask);
    return;
  }
  const SortScratch s = carve(tmp, static_cast<int64_t>(n), true);
  DevBuf scanScratch;
  // in -> ... -> out: the LAST pass writes 'out', the ones before alternate between it and the spare
  const int passes = (endBit - beginBit + 7) / 8;
  const uint64_t* src = in;
  for (int p = 0; p < passes; ++p) {
    uint64_t* dst = ((passes - 1 - p) & 1) ? s.spare : out;
    radixPass<false>(src, nullptr, dst, nullptr, static_cast<int64_t>(n), beginBit + p * 8, endBit, s.hist, s.offsets,
                     scanScratch);
    src = dst;
  }
  Runtime::get().sync();
}

}  // namespace vx
