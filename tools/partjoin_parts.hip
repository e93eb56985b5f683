// The three passes a radix-partitioned join of BASELINE config 5's one-GPU shape (200 M probe rows,
// 20 M-row table beyond every L2) would add or replace, each measured on its own - VERDICT r04 asked for
// numbers instead of an estimate before the idea is dropped or built:
//   A  partition pass: {key, row} records of the probe side to 256 partitions by hash bits through
//      LDS-sorted sub-tiles (the scheme of k_pp_scatter_fast / k_rp_scatter1_sorted);
//   B  partition-local probe: every partition's records against ITS 1/256 of the table (a 2 MiB slice of
//      16-byte slots: L2 resident while the partition is being probed), result {row, payload} written
//      next to the records;
//   C  back to probe-row order (HashProbe emits ascending probe rows): payload[row] = ... a random
//      8-byte store per probe row (100 % hit rate in config 5: sorting 200 M hits would cost more).
// Against them: the direct probe of the library on the same shape, k_join_probe 5.95 ms + k_emit 1.40 ms
// (profiles/r04_bench_c5_one_gpu.json). Build: hipcc --offload-arch=gfx950 -O3 tools/partjoin_parts.hip -o tools/partjoin_parts.bin
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(err_), __LINE__); exit(1); } } while (0)

__device__ __host__ inline uint64_t mix(uint64_t k) {
  k = (~k) + (k << 21); k ^= k >> 24; k = k + (k << 3) + (k << 8); k ^= k >> 14;
  k = k + (k << 2) + (k << 4); k ^= k >> 28; k = k + (k << 31);
  return k;
}

constexpr int kBins = 256;
constexpr int kSub = 8192;
constexpr int kThreads = 1024;
typedef unsigned long long U64x2 __attribute__((ext_vector_type(2)));

__global__ void k_fill(uint64_t* keys, int64_t n, uint64_t dimRows) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    keys[i] = (mix(static_cast<uint64_t>(i) * 0x9E3779B97F4A7C15ULL) % dimRows) * 7919 % (1ULL << 45);
  }
}

// A: records {key, row} to bin = top 8 bits of mix(key); bin b owns region [b * cap, ...) and a cursor.
__global__ __launch_bounds__(kThreads) void k_partition(const uint64_t* keys, int64_t n, U64x2* out, uint32_t* cursor, uint64_t cap) {
  __shared__ unsigned long long base[kBins];
  __shared__ uint32_t cnt[kBins], start[kBins];
  __shared__ U64x2 recs[kSub];
  __shared__ uint32_t waveTotals[kThreads / 64];
  const int tid = threadIdx.x;
  for (int i = tid; i < kBins; i += kThreads) {
    cnt[i] = 0;
  }
  __syncthreads();
  const int64_t numSub = (n + kSub - 1) / kSub;
  for (int64_t sub = blockIdx.x; sub < numSub; sub += gridDim.x) {
    const int64_t first = sub * kSub;
    uint64_t key[8];
    uint32_t bin[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t r = first + u * kThreads + tid;
      key[u] = __builtin_nontemporal_load(keys + (r < n ? r : n - 1));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t r = first + u * kThreads + tid;
      bin[u] = r < n ? static_cast<uint32_t>(mix(key[u]) >> 56) : 0xffffffffu;
      if (r < n) {
        atomicAdd(&cnt[bin[u]], 1u);
      }
    }
    __syncthreads();
    const uint32_t mine = tid < kBins ? cnt[tid] : 0;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(incl, off, 64);
      if ((tid & 63) >= off) {
        incl += o;
      }
    }
    if ((tid & 63) == 63) {
      waveTotals[tid >> 6] = incl;
    }
    __syncthreads();
    uint32_t run = incl - mine;
    for (int w = 0; w < (tid >> 6); ++w) {
      run += waveTotals[w];
    }
    if (tid < kBins) {
      start[tid] = run;
      cnt[tid] = run;
      if (mine) {
        base[tid] = static_cast<uint64_t>(tid) * cap + atomicAdd(&cursor[tid], mine);
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (bin[u] != 0xffffffffu) {
        const uint32_t pos = atomicAdd(&cnt[bin[u]], 1u);
        U64x2 rec;
        rec.x = key[u];
        rec.y = static_cast<uint64_t>(first + u * kThreads + tid);
        recs[pos] = rec;
      }
    }
    __syncthreads();
    const uint32_t total = static_cast<uint32_t>(first + kSub <= n ? kSub : n - first);
    for (uint32_t i = tid; i < total; i += kThreads) {
      const U64x2 rec = recs[i];
      const uint32_t b = static_cast<uint32_t>(mix(rec.x) >> 56);
      __builtin_nontemporal_store(rec, out + base[b] + (i - start[b]));
    }
    __syncthreads();
    if (tid < kBins) {
      cnt[tid] = 0;
    }
    __syncthreads();
  }
}

// B: partition p's records against slice p of the table (slots of 16 bytes, 2^17 per slice); {row, payload} out.
__global__ __launch_bounds__(1024) void k_local_probe(const U64x2* recs, const uint32_t* counts, uint64_t cap, const U64x2* table,
                                                       uint32_t slotsPerSlice, U64x2* out, int blocksPerPart) {
  // XCD-aware: workgroup ids go round-robin over the 8 XCDs, so the workgroups of ONE partition are
  // 8 ids apart (they share an XCD and its 4 MiB L2), and the partitions in flight on an XCD are few
  const int xcd = blockIdx.x & 7;
  const int local = blockIdx.x >> 3;
  const int part = (local / blocksPerPart) * 8 + xcd;
  const int sliceBlock = local % blocksPerPart;
  const uint32_t n = counts[part];
  const U64x2* in = recs + static_cast<uint64_t>(part) * cap;
  const U64x2* slice = table + static_cast<uint64_t>(part) * slotsPerSlice;
  U64x2* dst = out + static_cast<uint64_t>(part) * cap;
  for (uint32_t i = sliceBlock * 1024u * 4 + threadIdx.x; i < n; i += blocksPerPart * 1024u * 4) {
    U64x2 r[4], s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t j = i + u * 1024u < n ? i + u * 1024u : n - 1;
      r[u] = __builtin_nontemporal_load(in + j);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s[u] = slice[(mix(r[u].x) >> 20) & (slotsPerSlice - 1)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i + u * 1024u < n) {
        U64x2 o;
        o.x = r[u].y;
        o.y = s[u].y + (s[u].x == r[u].x ? 0 : 1);
        __builtin_nontemporal_store(o, dst + i + u * 1024u);
      }
    }
  }
}

// C: payload[row] = value for every {row, value}
__global__ __launch_bounds__(256) void k_write_back(const U64x2* pairs, const uint32_t* counts, uint64_t cap, uint64_t* payload) {
  const int part = blockIdx.y;
  const uint32_t n = counts[part];
  const U64x2* in = pairs + static_cast<uint64_t>(part) * cap;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const U64x2 p = __builtin_nontemporal_load(in + i);
    payload[p.x] = p.y;
  }
}

template <typename F>
float timeIt(F&& f, int iters) {
  hipEvent_t b, e;
  CK(hipEventCreate(&b));
  CK(hipEventCreate(&e));
  f();
  CK(hipEventRecord(b, 0));
  for (int i = 0; i < iters; ++i) {
    f();
  }
  CK(hipEventRecord(e, 0));
  CK(hipEventSynchronize(e));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, b, e));
  return ms / iters;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int64_t n = 200000000;
  const uint64_t dimRows = 20000000;
  const uint64_t cap = n / kBins + n / (2 * kBins) + 4096;
  uint64_t *keys, *payload;
  U64x2 *recs, *pairs, *table;
  uint32_t* cursor;
  const uint32_t slotsPerSlice = 1u << 17;   // 2 MiB per partition: 32 M slots for 20 M keys
  CK(hipMalloc(&keys, n * 8));
  CK(hipMalloc(&payload, n * 8));
  CK(hipMalloc(&recs, cap * kBins * 16));
  CK(hipMalloc(&pairs, cap * kBins * 16));
  CK(hipMalloc(&table, static_cast<uint64_t>(slotsPerSlice) * kBins * 16));
  CK(hipMalloc(&cursor, kBins * 4));
  CK(hipMemset(table, 0x11, static_cast<uint64_t>(slotsPerSlice) * kBins * 16));
  hipLaunchKernelGGL(k_fill, dim3(cus * 8), dim3(256), 0, 0, keys, n, dimRows);
  CK(hipDeviceSynchronize());
  printf("# %d CUs; %lld probe rows, %d partitions, %u 16-byte slots per partition slice\n", cus, (long long)n, kBins, slotsPerSlice);
  const float a = timeIt([&] {
    CK(hipMemsetAsync(cursor, 0, kBins * 4, 0));
    hipLaunchKernelGGL(k_partition, dim3(cus), dim3(kThreads), 0, 0, keys, n, recs, cursor, cap);
  }, 5);
  printf("A partition pass (8 B key in, 16 B {key, row} out, LDS-sorted sub-tiles of 8192): %.3f ms = %.0f GB/s\n", a, n * 24.0 / (a * 1e-3) / 1e9);
  for (int bpp : {8, 16, 32, 64}) {
    const float b = timeIt([&] {
      hipLaunchKernelGGL(k_local_probe, dim3(kBins * bpp), dim3(1024), 0, 0, recs, cursor, cap, table, slotsPerSlice, pairs, bpp);
    }, 5);
    printf("B partition-local probe, %2d workgroups per partition, all on one XCD (16 B record in, one 16 B slot of a 2 MiB slice, 16 B out): %.3f ms = %.1f G probes/s\n",
           bpp, b, n / (b * 1e-3) / 1e9);
  }
  const float c = timeIt([&] {
    hipLaunchKernelGGL(k_write_back, dim3(64, kBins), dim3(256), 0, 0, pairs, cursor, cap, payload);
  }, 5);
  printf("C back to probe-row order (payload[row] = value, one random 8 B store per probe row): %.3f ms = %.1f G rows/s\n", c, n / (c * 1e-3) / 1e9);
  printf("(instead of C: two more passes like A over {row, value} records, by row bits, then a linear write: >= 2 x A + 0.5 ms)\n");
  printf("A + B(best) + C vs the direct probe's k_join_probe + k_emit = 5.95 + 1.40 ms on this shape (profiles/r04_bench_c5_one_gpu.json)\n");
  return 0;
}
