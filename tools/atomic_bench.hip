// Micro-benchmark behind the high-cardinality aggregation design (DESIGN.md "config 4"): how many
// f64 atomic adds per second does the chip serve on a table in HBM, as a function of
//   - the scope of the atomic (agent = what any kernel may use across XCDs; workgroup = executed by
//     the XCD's own L2, only sound when every workgroup touching a line runs on the same XCD),
//   - whether the workgroups of one XCD confine themselves to one slice of the table small
//     enough for that XCD's 4 MiB L2 ("XCD-affine": blockIdx % 8 owns slices s with s % 8 ==
//     blockIdx % 8, what a radix partition pass in front of the fold buys).
// Every operation streams an 8-byte key in and adds 1.0 at a random position of its slice; the
// table is summed afterwards, so lost updates show up as a wrong total.
// Build: hipcc --offload-arch=gfx950 -O3 tools/atomic_bench.hip -o tools/atomic_bench.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ inline uint64_t mix(uint64_t k) {
  k = (~k) + (k << 21); k ^= k >> 24; k = k + (k << 3) + (k << 8); k ^= k >> 14;
  k = k + (k << 2) + (k << 4); k ^= k >> 28; k = k + (k << 31);
  return k;
}

// MODE 0: agent-scope atomic, 1: workgroup-scope atomic, 2: plain read-modify-write (racy upper bound).
// sliceSlots: f64 slots per slice; numSlices slices; AFFINE: workgroup b works on slices
// (b % 8) + 8 * j in turn, a pass over 'perSlice' keys each, together with the other workgroups of
// its XCD; otherwise positions are random over the whole table.
template <int MODE, bool AFFINE>
__global__ __launch_bounds__(256) void k_atomic(const uint64_t* keys, int64_t n, double* table, uint64_t sliceSlots,
                                                uint64_t numSlices) {
  const uint64_t totalSlots = sliceSlots * numSlices;
  if (!AFFINE) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
      double* p = table + mix(keys[i]) % totalSlots;
      if (MODE == 0) {
        __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (MODE == 1) {
        __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        *p += 1.0;
      }
    }
    return;
  }
  // keys are consumed in numSlices equal runs: run s feeds slice s (as if a partition pass had
  // grouped them); the workgroups of XCD x = blockIdx % 8 share the slices s % 8 == x.
  const int64_t perSlice = n / static_cast<int64_t>(numSlices);
  const int xcd = blockIdx.x & 7;
  const int member = blockIdx.x >> 3;          // index among the workgroups of this XCD
  const int members = (gridDim.x + 7 - xcd) >> 3;
  for (uint64_t s = xcd; s < numSlices; s += 8) {
    double* slice = table + s * sliceSlots;
    const int64_t begin = static_cast<int64_t>(s) * perSlice;
    for (int64_t i = begin + member * 256 + threadIdx.x; i < begin + perSlice; i += static_cast<int64_t>(members) * 256) {
      double* p = slice + mix(keys[i]) % sliceSlots;
      if (MODE == 0) {
        __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (MODE == 1) {
        __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        *p += 1.0;
      }
    }
  }
}

__global__ void k_fill_keys(uint64_t* keys, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    keys[i] = mix(static_cast<uint64_t>(i) * 0x9E3779B97F4A7C15ULL + 1);
  }
}

__global__ void k_sum(const double* table, uint64_t slots, double* out) {
  double s = 0;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < slots; i += stride) {
    s += table[i];
  }
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out, s);
  }
}

template <int MODE, bool AFFINE>
int run(const char* label, const uint64_t* keys, int64_t n, double* table, uint64_t sliceSlots, uint64_t numSlices,
        double* dSum, int grid) {
  const uint64_t slots = sliceSlots * numSlices;
  CK(hipMemset(table, 0, slots * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int64_t used = AFFINE ? (n / static_cast<int64_t>(numSlices)) * static_cast<int64_t>(numSlices) : n;
  k_atomic<MODE, AFFINE><<<grid, 256>>>(keys, n, table, sliceSlots, numSlices);  // warm
  CK(hipDeviceSynchronize());
  CK(hipMemset(table, 0, slots * 8));
  CK(hipEventRecord(e0));
  k_atomic<MODE, AFFINE><<<grid, 256>>>(keys, n, table, sliceSlots, numSlices);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemset(dSum, 0, 8));
  k_sum<<<2048, 256>>>(table, slots, dSum);
  double total = 0;
  CK(hipMemcpy(&total, dSum, 8, hipMemcpyDeviceToHost));
  printf("%-34s table %8.1f MiB slices of %7.2f MiB grid %5d: %8.3f ms %8.1f G adds/s  sum %s (%.0f of %lld)\n", label,
         slots * 8.0 / (1 << 20), sliceSlots * 8.0 / (1 << 20), grid, ms, used / (ms * 1e-3) / 1e9,
         total == static_cast<double>(used) ? "exact" : "LOST UPDATES", total, static_cast<long long>(used));
  return 0;
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 400000000LL;
  uint64_t* keys;
  double* table;
  double* dSum;
  const uint64_t maxBytes = 4ULL << 30;
  CK(hipMalloc(&keys, n * 8));
  CK(hipMalloc(&table, maxBytes));
  CK(hipMalloc(&dSum, 8));
  k_fill_keys<<<2048, 256>>>(keys, n);
  CK(hipDeviceSynchronize());
  const int grid = 2048;
  // random over the whole table: what k_agg_global does per DOUBLE-sum word
  for (uint64_t mib : {64ULL, 1024ULL, 4096ULL}) {
    const uint64_t slots = mib << 17;
    if (run<0, false>("agent scope, whole table", keys, n, table, slots, 1, dSum, grid)) return 1;
    if (run<2, false>("plain RMW (racy), whole table", keys, n, table, slots, 1, dSum, grid)) return 1;
  }
  // XCD-affine slices of a 3.2 GB-class table
  for (double sliceMib : {0.5, 1.0, 2.0, 4.0, 16.0}) {
    const uint64_t sliceSlots = static_cast<uint64_t>(sliceMib * (1 << 17));
    const uint64_t numSlices = (3ULL << 30) / (sliceSlots * 8) / 8 * 8;
    if (run<0, true>("agent scope, XCD-affine slices", keys, n, table, sliceSlots, numSlices, dSum, grid)) return 1;
    if (run<1, true>("workgroup scope, XCD-affine slices", keys, n, table, sliceSlots, numSlices, dSum, grid)) return 1;
    if (run<2, true>("plain RMW (racy), XCD-affine", keys, n, table, sliceSlots, numSlices, dSum, grid)) return 1;
  }
  return 0;
}
