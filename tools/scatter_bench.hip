// Micro-benchmark behind the radix-partitioned aggregation (DESIGN.md "high cardinality"): what
// does one partition pass cost — 8-byte key + 8-byte value in, one 16-byte record out to one of
// BINS buckets — as a function of HOW the records are written?
//   direct : every lane stores its record at base[tile][bin] + LDS cursor (k_rp_scatter1 today)
//   sorted : sub-tiles of SUB records are counting-sorted by bin in LDS first, so the records of
//            a bin leave as runs of consecutive lanes (k_pp_scatter's scheme)
//   window : the output is laid out [super-tile][bin][tile] instead of [bin][tile]: a workgroup's
//            BINS write streams then fall into one window of n / S records instead of the whole
//            buffer (TLB reach, DRAM page locality)
// and of the workgroup shape (threads, rows per lane in flight). Rates are (read + write) bytes
// over the kernel time of the scatter alone; the histogram + scan in front are not timed.
// Build: hipcc --offload-arch=gfx950 -O3 tools/scatter_bench.hip -o tools/scatter_bench.bin
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kMaxBins = 1024;

__device__ inline uint64_t mix(uint64_t k) {
  k = (~k) + (k << 21); k ^= k >> 24; k = k + (k << 3) + (k << 8); k ^= k >> 14;
  k = k + (k << 2) + (k << 4); k ^= k >> 28; k = k + (k << 31);
  return k;
}

__global__ void k_fill(uint64_t* keys, uint64_t* vals, int64_t n, uint64_t range) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    keys[i] = mix(static_cast<uint64_t>(i) * 0x9E3779B97F4A7C15ULL) % range;
    vals[i] = static_cast<uint64_t>(i);
  }
}

struct Args {
  const uint64_t* keys;
  const uint64_t* vals;
  int64_t n;
  int64_t tileRows;
  int64_t numTiles;
  int32_t bins;
  int32_t shift;        // bin = key >> shift
  int64_t tilesPerSuper;  // window layout: tiles per super-tile (0 = plain [bin][tile])
  uint32_t* hist;       // cell(bin, tile)
  const uint64_t* offsets;
  ulonglong2* out;
};

__device__ inline int64_t cellOf(const Args& a, int32_t bin, int64_t tile) {
  if (a.tilesPerSuper == 0) {
    return static_cast<int64_t>(bin) * a.numTiles + tile;
  }
  const int64_t s = tile / a.tilesPerSuper, t = tile % a.tilesPerSuper;
  // the last super-tile may be short: its cells still use tilesPerSuper slots
  return (s * a.bins + bin) * a.tilesPerSuper + t;
}

template <int T>
__global__ __launch_bounds__(T) void k_hist(Args a) {
  __shared__ uint32_t hist[kMaxBins];
  for (int64_t tile = blockIdx.x; tile < a.numTiles; tile += gridDim.x) {
    for (int i = threadIdx.x; i < a.bins; i += T) {
      hist[i] = 0;
    }
    __syncthreads();
    const int64_t begin = tile * a.tileRows;
    const int64_t end = begin + a.tileRows < a.n ? begin + a.tileRows : a.n;
    for (int64_t r = begin + threadIdx.x; r < end; r += T) {
      atomicAdd(&hist[a.keys[r] >> a.shift], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.bins; i += T) {
      a.hist[cellOf(a, i, tile)] = hist[i];
    }
    __syncthreads();
  }
}

// direct: T threads, U rows per lane in flight
template <int T, int U, bool NT>
__global__ __launch_bounds__(T) void k_direct(Args a) {
  __shared__ unsigned long long binBase[kMaxBins];
  __shared__ uint32_t cursor[kMaxBins];
  for (int64_t tile = blockIdx.x; tile < a.numTiles; tile += gridDim.x) {
    for (int i = threadIdx.x; i < a.bins; i += T) {
      binBase[i] = a.offsets[cellOf(a, i, tile)];
      cursor[i] = 0;
    }
    __syncthreads();
    const int64_t begin = tile * a.tileRows;
    const int64_t end = begin + a.tileRows < a.n ? begin + a.tileRows : a.n;
    for (int64_t base = begin; base < end; base += static_cast<int64_t>(U) * T) {
      uint64_t k[U], v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = base + u * T + threadIdx.x;
        k[u] = r < end ? a.keys[r] : 0;
        v[u] = r < end ? a.vals[r] : 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = base + u * T + threadIdx.x;
        if (r < end) {
          const uint32_t bin = static_cast<uint32_t>(k[u] >> a.shift);
          const unsigned long long pos = binBase[bin] + atomicAdd(&cursor[bin], 1u);
          if (NT) {
            __builtin_nontemporal_store(k[u], reinterpret_cast<uint64_t*>(a.out + pos));
            __builtin_nontemporal_store(v[u], reinterpret_cast<uint64_t*>(a.out + pos) + 1);
          } else {
            a.out[pos] = make_ulonglong2(k[u], v[u]);
          }
        }
      }
    }
    __syncthreads();
  }
}

// sorted: sub-tiles of SUB = T * U records counting-sorted by bin in LDS, then written as runs
template <int T, int U>
__global__ __launch_bounds__(T) void k_sorted(Args a) {
  constexpr int SUB = T * U;
  __shared__ unsigned long long binBase[kMaxBins];
  __shared__ uint32_t cnt[kMaxBins];
  __shared__ uint32_t start[kMaxBins];
  __shared__ ulonglong2 recs[SUB];
  __shared__ uint16_t binOf[SUB];
  __shared__ uint32_t waveTotals[T / 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int PER = (kMaxBins + T - 1) / T;
  for (int64_t tile = blockIdx.x; tile < a.numTiles; tile += gridDim.x) {
    for (int i = tid; i < a.bins; i += T) {
      binBase[i] = a.offsets[cellOf(a, i, tile)];
    }
    const int64_t begin = tile * a.tileRows;
    const int64_t end = begin + a.tileRows < a.n ? begin + a.tileRows : a.n;
    for (int64_t base = begin; base < end; base += SUB) {
      for (int i = tid; i < a.bins; i += T) {
        cnt[i] = 0;
      }
      __syncthreads();
      uint64_t k[U], v[U];
      uint32_t bin[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = base + u * T + tid;
        k[u] = r < end ? a.keys[r] : 0;
        v[u] = r < end ? a.vals[r] : 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = base + u * T + tid;
        bin[u] = 0xffffffffu;
        if (r < end) {
          bin[u] = static_cast<uint32_t>(k[u] >> a.shift);
          atomicAdd(&cnt[bin[u]], 1u);
        }
      }
      __syncthreads();
      uint32_t mine[PER];
      uint32_t sum = 0;
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int b = tid * PER + q;
        mine[q] = b < a.bins ? cnt[b] : 0;
        sum += mine[q];
      }
      uint32_t incl = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) {
          incl += o;
        }
      }
      if (lane == 63) {
        waveTotals[wave] = incl;
      }
      __syncthreads();
      uint32_t run = incl - sum;
      for (int w = 0; w < wave; ++w) {
        run += waveTotals[w];
      }
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int b = tid * PER + q;
        if (b < a.bins) {
          start[b] = run;
          cnt[b] = run;
        }
        run += mine[q];
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (bin[u] != 0xffffffffu) {
          const uint32_t pos = atomicAdd(&cnt[bin[u]], 1u);
          recs[pos] = make_ulonglong2(k[u], v[u]);
          binOf[pos] = static_cast<uint16_t>(bin[u]);
        }
      }
      __syncthreads();
      uint32_t total = 0;
      for (int w = 0; w < T / 64; ++w) {
        total += waveTotals[w];
      }
      for (uint32_t i = tid; i < total; i += T) {
        const uint32_t b = binOf[i];
        a.out[binBase[b] + (i - start[b])] = recs[i];
      }
      __syncthreads();
      for (int b = tid; b < a.bins; b += T) {
        binBase[b] += cnt[b] - start[b];
      }
      __syncthreads();
    }
  }
}

__global__ void k_copy(const ulonglong2* in, ulonglong2* out, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    out[i] = in[i];
  }
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 500000000LL;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint64_t *keys, *vals;
  ulonglong2* out;
  CK(hipMalloc(&keys, n * 8));
  CK(hipMalloc(&vals, n * 8));
  CK(hipMalloc(&out, n * 16 + (64 << 20)));
  const uint64_t range = 200000000ULL;  // config 4: 2 x 10^8 group rows
  k_fill<<<cus * 8, 256>>>(keys, vals, n, range);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timeIt = [&](auto&& launch, int reps = 3) {
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) {
      launch();
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
  };
  {
    const float ms = timeIt([&] { k_copy<<<cus * 8, 256>>>(reinterpret_cast<const ulonglong2*>(keys), out, n / 2); });
    printf("copy 16-byte words: %.3f ms  %.2f TB/s (read + write)\n", ms, n * 16.0 / ms / 1e9);
  }
  uint32_t* hist;
  uint64_t* offsets;
  void* scanTmp = nullptr;
  size_t scanBytes = 0;
  const int64_t maxCells = 1LL << 27;
  CK(hipMalloc(&hist, maxCells * 4));
  CK(hipMalloc(&offsets, (maxCells + 1) * 8));
  CK(rocprim::exclusive_scan(nullptr, scanBytes, hist, offsets, 0ULL, maxCells, rocprim::plus<uint64_t>()));
  CK(hipMalloc(&scanTmp, scanBytes));
  struct Case {
    const char* name;
    int bins;
    int64_t tileRows;
    int superTiles;  // 0 = plain layout
    int variant;     // 0 direct 1024x8, 1 direct NT, 2 direct 512x8, 3 direct 256x16, 4 sorted 1024x4, 5 sorted 512x8, 6 direct 1024x16
  };
  std::vector<Case> cases;
  for (int bins : {64, 256, 512}) {
    for (int super : {0, 64}) {
      for (int variant : {0, 1, 2, 3, 4, 5, 6}) {
        cases.push_back({"", bins, 0, super, variant});
      }
    }
  }
  const char* vname[] = {"direct 1024thr x8", "direct 1024thr x8 nontemporal", "direct 512thr x8", "direct 256thr x16",
                         "sorted 1024thr sub4096", "sorted 512thr sub4096", "direct 1024thr x16"};
  for (auto& c : cases) {
    Args a{};
    a.keys = keys;
    a.vals = vals;
    a.n = n;
    a.bins = c.bins;
    int shift = 0;
    while ((range >> shift) > static_cast<uint64_t>(c.bins)) {
      ++shift;
    }
    a.shift = shift;
    a.bins = static_cast<int32_t>(((range - 1) >> shift) + 1);
    int64_t tileRows = std::max<int64_t>(32768, (n + 2047) / 2048);
    tileRows = (tileRows + 4095) & ~4095LL;
    a.tileRows = tileRows;
    a.numTiles = (n + tileRows - 1) / tileRows;
    a.tilesPerSuper = c.superTiles ? (a.numTiles + c.superTiles - 1) / c.superTiles : 0;
    const int64_t cells = c.superTiles ? static_cast<int64_t>(c.superTiles) * a.bins * a.tilesPerSuper
                                       : static_cast<int64_t>(a.bins) * a.numTiles;
    if (cells > maxCells) {
      continue;
    }
    a.hist = hist;
    a.offsets = offsets;
    a.out = out;
    CK(hipMemset(hist, 0, cells * 4));
    const int grid = static_cast<int>(std::min<int64_t>(a.numTiles, cus * 2));
    k_hist<1024><<<grid, 1024>>>(a);
    CK(rocprim::exclusive_scan(scanTmp, scanBytes, hist, offsets, 0ULL, cells, rocprim::plus<uint64_t>()));
    CK(hipDeviceSynchronize());
    float ms = 0;
    switch (c.variant) {
      case 0: ms = timeIt([&] { k_direct<1024, 8, false><<<grid, 1024>>>(a); }); break;
      case 1: ms = timeIt([&] { k_direct<1024, 8, true><<<grid, 1024>>>(a); }); break;
      case 2: ms = timeIt([&] { k_direct<512, 8, false><<<std::min<int64_t>(a.numTiles, cus * 4), 512>>>(a); }); break;
      case 3: ms = timeIt([&] { k_direct<256, 16, false><<<std::min<int64_t>(a.numTiles, cus * 8), 256>>>(a); }); break;
      case 4: ms = timeIt([&] { k_sorted<1024, 4><<<grid, 1024>>>(a); }); break;
      case 5: ms = timeIt([&] { k_sorted<512, 8><<<grid, 512>>>(a); }); break;
      default: ms = timeIt([&] { k_direct<1024, 16, false><<<grid, 1024>>>(a); }); break;
    }
    // spot check: every output record sits in its bin's range (last launch's output)
    printf("bins %4d  layout %-22s %-30s %8.3f ms  %.2f TB/s\n", a.bins, c.superTiles ? "[super 64][bin][tile]" : "[bin][tile]",
           vname[c.variant], ms, n * 32.0 / ms / 1e9);
    fflush(stdout);
  }
  return 0;
}
