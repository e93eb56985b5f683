// Micro-benchmark behind the join-probe design (DESIGN.md "probe"): how many dependent random
// reads per second does the chip serve, as a function of the size of the structure probed
// (L2 4 MiB per XCD, Infinity Cache 256 MiB, HBM) and of whether the workgroups of one XCD
// confine themselves to one slice of it ("XCD-affine": blockIdx % 8 owns slice blockIdx % 8,
// which is what a radix-partitioned probe buys). Every probe streams an 8-byte key in, reads
// W bytes at a random position of the table and writes 4 bytes out — the shape of
// k_join_probe. Build: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o tools/gather_bench.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ inline uint64_t mix(uint64_t k) {
  k = (~k) + (k << 21); k ^= k >> 24; k = k + (k << 3) + (k << 8); k ^= k >> 14;
  k = k + (k << 2) + (k << 4); k ^= k >> 28; k = k + (k << 31);
  return k;
}

// W = bytes read per probe (4, 8, 16). AFFINE: the table is cut into 8 slices, workgroups with
// blockIdx % 8 == x only touch slice x. U probes per lane in flight.
template <int W, bool AFFINE, int U>
__global__ __launch_bounds__(256) void k_gather(const uint64_t* keys, int64_t n, const uint32_t* table,
                                                uint64_t tableWords, uint32_t* out) {
  const uint64_t sliceWords = AFFINE ? tableWords / 8 : tableWords;
  const uint64_t sliceBase = AFFINE ? (blockIdx.x & 7) * sliceWords : 0;
  const uint64_t unitWords = W / 4;
  const uint64_t units = sliceWords / unitWords;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256 * U;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * 256 * U; base < n; base += stride) {
    uint64_t k[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * 256 + threadIdx.x;
      k[u] = keys[i < n ? i : n - 1];
    }
    uint32_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t pos = sliceBase + (mix(k[u]) % units) * unitWords;
      if (W == 4) {
        v[u] = table[pos];
      } else if (W == 8) {
        const uint2 t = *reinterpret_cast<const uint2*>(table + pos);
        v[u] = t.x ^ t.y;
      } else {
        const uint4 t = *reinterpret_cast<const uint4*>(table + pos);
        v[u] = t.x ^ t.y ^ t.z ^ t.w;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * 256 + threadIdx.x;
      if (i < n) {
        out[i] = v[u];
      }
    }
  }
}

// Bit-test flavour (the presence bitmap of the array-mode join): 1 bit per possible key.
template <bool AFFINE, int U>
__global__ __launch_bounds__(256) void k_bits(const uint64_t* keys, int64_t n, const uint32_t* table,
                                              uint64_t tableWords, uint32_t* out) {
  const uint64_t sliceWords = AFFINE ? tableWords / 8 : tableWords;
  const uint64_t sliceBase = AFFINE ? (blockIdx.x & 7) * sliceWords : 0;
  const uint64_t bits = sliceWords * 32;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256 * U;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * 256 * U; base < n; base += stride) {
    uint64_t k[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * 256 + threadIdx.x;
      k[u] = keys[i < n ? i : n - 1];
    }
    uint32_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t b = mix(k[u]) % bits;
      v[u] = (table[sliceBase + (b >> 5)] >> (b & 31)) & 1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * 256 + threadIdx.x;
      if (i < n) {
        out[i] = v[u];
      }
    }
  }
}

// The wave-cooperative bucket of the F14-style design (one 64-byte line of 64 one-byte tags per
// bucket, all 64 lanes compare it with one ballot): every probe costs the WAVE one line read,
// so a wave has U probes in flight instead of 64 * U. Upper bound of that design: the tag line
// only, no key confirmation behind it.
template <bool AFFINE, int U>
__global__ __launch_bounds__(256) void k_tagline(const uint64_t* keys, int64_t n, const uint32_t* table,
                                                 uint64_t tableWords, uint32_t* out) {
  const uint8_t* tags = reinterpret_cast<const uint8_t*>(table);
  const uint64_t sliceBytes = (AFFINE ? tableWords / 8 : tableWords) * 4;
  const uint64_t sliceBase = AFFINE ? (blockIdx.x & 7) * sliceBytes : 0;
  const uint64_t lines = sliceBytes / 64;
  const int ln = threadIdx.x & 63;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
  const int64_t waves = static_cast<int64_t>(gridDim.x) * 4;
  for (int64_t base = wave * 64; base < n; base += waves * 64) {
    const uint64_t mine = keys[base + ln < n ? base + ln : n - 1];
    uint32_t result = 0;
    for (int j = 0; j < 64; j += U) {
      uint64_t h[U];
      uint8_t t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t lo = __shfl(static_cast<uint32_t>(mine), j + u, 64);
        const uint32_t hi = __shfl(static_cast<uint32_t>(mine >> 32), j + u, 64);
        h[u] = mix((static_cast<uint64_t>(hi) << 32) | lo);
        t[u] = tags[sliceBase + (h[u] % lines) * 64 + ln];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t m = __ballot(t[u] == static_cast<uint8_t>(h[u] >> 56));
        if (ln == j + u) {
          result = static_cast<uint32_t>(m) ^ static_cast<uint32_t>(m >> 32) ^ t[u];
        }
      }
    }
    if (base + ln < n) {
      out[base + ln] = result;
    }
  }
}

// Random 16-byte record scatter into B open bins (the partition pass): where do writes top out?
__global__ __launch_bounds__(256) void k_stream_copy(const uint4* in, uint4* out, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
    out[i] = in[i];
  }
}

int main(int argc, char** argv) {
  const int64_t n = 1LL << 27;
  uint64_t* keys;
  uint32_t* out;
  uint32_t* table;
  const uint64_t maxBytes = 4ULL << 30;
  CK(hipMalloc(&keys, n * 8));
  CK(hipMalloc(&out, n * 4));
  CK(hipMalloc(&table, maxBytes));
  CK(hipMemset(table, 0x5a, maxBytes));
  std::vector<uint64_t> hk(1 << 22);
  uint64_t x = 88172645463325252ULL;
  for (auto& k : hk) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; k = x; }
  for (int64_t off = 0; off < n; off += (int64_t)hk.size()) {
    for (auto& k : hk) k += 0x9E3779B97F4A7C15ULL;   // different keys in every block of 4 M
    CK(hipMemcpy(keys + off, hk.data(), hk.size() * 8, hipMemcpyHostToDevice));
  }
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  auto run = [&](const char* name, auto kern, uint64_t bytes, int grid) {
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, keys, n, table, bytes / 4, out);
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, keys, n, table, bytes / 4, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 3;
    printf("%-26s table %8.1f MiB grid %5d: %7.3f ms  %7.1f G probes/s  %6.0f GB/s at 12 B/probe\n", name,
           bytes / 1048576.0, grid, ms, n / ms / 1e6, n * 12.0 / ms / 1e6);
    fflush(stdout);
  };
  if (argc > 1 && std::string(argv[1]) == "tagline") {
    // wave-cooperative tag lines vs one lane per probe, at an L2-sized and an HBM-sized table
    for (uint64_t s : {16ULL << 20, 512ULL << 20}) {
      run("gather16 U4 (lane/probe)", (k_gather<16, false, 4>), s, 2048);
      run("tagline  U4 (wave/probe)", (k_tagline<false, 4>), s, 2048);
      run("tagline  U8 (wave/probe)", (k_tagline<false, 8>), s, 2048);
      run("tagline  U16 (wave/probe)", (k_tagline<false, 16>), s, 2048);
      run("tagline  U8 xcd-affine", (k_tagline<true, 8>), s, 2048);
    }
    return 0;
  }
  const uint64_t sizes[] = {1ULL << 20, 4ULL << 20, 16ULL << 20, 32ULL << 20, 64ULL << 20, 128ULL << 20,
                            256ULL << 20, 512ULL << 20, 1ULL << 30, 4ULL << 30};
  for (uint64_t s : sizes) {
    run("gather4  U4", (k_gather<4, false, 4>), s, 2048);
    run("gather4  U4 xcd-affine", (k_gather<4, true, 4>), s, 2048);
    run("gather16 U4", (k_gather<16, false, 4>), s, 2048);
    run("gather16 U4 xcd-affine", (k_gather<16, true, 4>), s, 2048);
    run("bits     U4", (k_bits<false, 4>), s, 2048);
    run("bits     U4 xcd-affine", (k_bits<true, 4>), s, 2048);
  }
  // loads in flight and grid size at two interesting sizes
  for (uint64_t s : {16ULL << 20, 512ULL << 20}) {
    run("gather16 U1", (k_gather<16, false, 1>), s, 2048);
    run("gather16 U2", (k_gather<16, false, 2>), s, 2048);
    run("gather16 U8", (k_gather<16, false, 8>), s, 2048);
    run("gather16 U8 xcd-affine", (k_gather<16, true, 8>), s, 2048);
    run("gather16 U4 grid 4096", (k_gather<16, false, 4>), s, 4096);
    run("gather16 U4 grid 1024", (k_gather<16, false, 4>), s, 1024);
  }
  // streaming copy for scale (read + write)
  {
    const int64_t m = 1LL << 27;  // 2 GiB of uint4
    uint4* src = reinterpret_cast<uint4*>(table);
    uint4* dst = src + m;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_stream_copy, dim3(4096), dim3(256), 0, 0, src, dst, m);
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_stream_copy, dim3(4096), dim3(256), 0, 0, src, dst, m);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 3;
    printf("stream copy 2 GiB -> 2 GiB: %7.3f ms  %6.0f GB/s (read + write)\n", ms, 2.0 * m * 16 / ms / 1e6);
  }
  return 0;
}
