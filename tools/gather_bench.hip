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


// ---- bucket shapes for the generic (kHash) mode: key confirmation included -----------------------
// All three read, per probe, (1) the slot / tag group at a hashed position, (2) the row id it
// names (dependent), (3) the 16-byte key image of that row (dependent) and compare it — what a
// hit costs when keys have no 64-bit normalized form (strings, wide key sets). The table is not
// really built: the "matching" slot of a bucket and the row it names are derived from the hash,
// which keeps the access pattern and the dependency chain of a real probe.
//   slotkey : the library's layout — one 8-byte {tag32, row} slot per lane, then the key image
//   tagswar : F14-shaped bucket (16 one-byte tags + 16 four-byte row ids in one 128-byte line,
//             exec/HashTable.h:897-930), ONE LANE per probe: 16-byte tag load, SWAR compare
//   taggroup<G>: the same bucket, G lanes per probe (8 or 16): every lane loads 16 / G tag bytes,
//             a ballot restricted to the group finds the match, one lane fetches id and key
template <int U>
__global__ __launch_bounds__(256) void k_slotkey(const uint64_t* keys, int64_t n, const uint32_t* table,
                                                 uint64_t tableWords, uint32_t* out) {
  const uint64_t slotBytes = tableWords * 4 / 3;              // a third of the area: slots; the rest: key images
  const uint64_t numSlots = slotBytes / 8;
  const uint64_t numKeys = (tableWords * 4 - slotBytes) / 16;
  const uint64_t* slots = reinterpret_cast<const uint64_t*>(table);
  const uint4* images = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(table) + slotBytes);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256 * U;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * 256 * U; base < n; base += stride) {
    uint64_t h[U], s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * 256 + threadIdx.x;
      h[u] = mix(keys[i < n ? i : n - 1]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s[u] = slots[h[u] % numSlots];
    }
    uint4 img[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      img[u] = images[mix(s[u] ^ h[u]) % numKeys];  // row named by the slot
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * 256 + threadIdx.x;
      if (i < n) {
        out[i] = (img[u].x == static_cast<uint32_t>(h[u])) + img[u].y + img[u].z + img[u].w;
      }
    }
  }
}

template <int U>
__global__ __launch_bounds__(256) void k_tagswar(const uint64_t* keys, int64_t n, const uint32_t* table,
                                                 uint64_t tableWords, uint32_t* out) {
  const uint64_t bucketBytes = tableWords * 4 / 3;
  const uint64_t numBuckets = bucketBytes / 128;
  const uint64_t numKeys = (tableWords * 4 - bucketBytes) / 16;
  const char* buckets = reinterpret_cast<const char*>(table);
  const uint4* images = reinterpret_cast<const uint4*>(buckets + bucketBytes);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256 * U;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * 256 * U; base < n; base += stride) {
    uint64_t h[U];
    uint4 tags[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * 256 + threadIdx.x;
      h[u] = mix(keys[i < n ? i : n - 1]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      tags[u] = *reinterpret_cast<const uint4*>(buckets + (h[u] % numBuckets) * 128);
    }
    uint32_t id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // SWAR: which of the 16 tag bytes equal the probe's tag (the table holds 0x5a everywhere, so
      // every byte "matches"; the hash picks which match is taken, as a real probe takes the first)
      const uint32_t want = 0x5a5a5a5au;
      uint32_t hits = 0;
      const uint32_t w[4] = {tags[u].x, tags[u].y, tags[u].z, tags[u].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t x = w[q] ^ want;
        const uint32_t zero = (x - 0x01010101u) & ~x & 0x80808080u;   // 0x80 in every byte that is zero
        hits |= ((zero >> 7) & 1) << (4 * q) | ((zero >> 15) & 1) << (4 * q + 1) | ((zero >> 23) & 1) << (4 * q + 2) |
            ((zero >> 31) & 1) << (4 * q + 3);
      }
      const uint32_t slot = (hits ? static_cast<uint32_t>(h[u] >> 20) : 0u) & 15u;
      id[u] = *reinterpret_cast<const uint32_t*>(buckets + (h[u] % numBuckets) * 128 + 16 + slot * 4);
    }
    uint4 img[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      img[u] = images[mix(id[u] ^ h[u]) % numKeys];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * 256 + threadIdx.x;
      if (i < n) {
        out[i] = (img[u].x == static_cast<uint32_t>(h[u])) + img[u].y + img[u].z + img[u].w;
      }
    }
  }
}

template <int G, int U>
__global__ __launch_bounds__(256) void k_taggroup(const uint64_t* keys, int64_t n, const uint32_t* table,
                                                  uint64_t tableWords, uint32_t* out) {
  constexpr int kPerWave = 64 / G;            // probes a wave works on at once
  const uint64_t bucketBytes = tableWords * 4 / 3;
  const uint64_t numBuckets = bucketBytes / 128;
  const uint64_t numKeys = (tableWords * 4 - bucketBytes) / 16;
  const char* buckets = reinterpret_cast<const char*>(table);
  const uint4* images = reinterpret_cast<const uint4*>(buckets + bucketBytes);
  const int ln = threadIdx.x & 63;
  const int group = ln / G, inGroup = ln % G;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
  const int64_t waves = static_cast<int64_t>(gridDim.x) * 4;
  for (int64_t base = wave * 64; base < n; base += waves * 64) {
    const uint64_t mine = keys[base + ln < n ? base + ln : n - 1];
    uint32_t result = 0;
    for (int j = 0; j < 64; j += kPerWave * U) {
      uint64_t h[U];
      uint32_t t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int src = j + u * kPerWave + group;   // the probe this group serves
        const uint32_t lo = __shfl(static_cast<uint32_t>(mine), src, 64);
        const uint32_t hi = __shfl(static_cast<uint32_t>(mine >> 32), src, 64);
        h[u] = mix((static_cast<uint64_t>(hi) << 32) | lo);
        const char* b = buckets + (h[u] % numBuckets) * 128;
        t[u] = G == 16 ? static_cast<uint32_t>(*reinterpret_cast<const uint8_t*>(b + inGroup))
                       : static_cast<uint32_t>(*reinterpret_cast<const uint16_t*>(b + inGroup * 2));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool match = G == 16 ? t[u] == 0x5au : ((t[u] & 0xff) == 0x5au || (t[u] >> 8) == 0x5au);
        const uint64_t all = __ballot(match);
        const uint32_t mineMask = static_cast<uint32_t>(all >> (group * G)) & ((1u << G) - 1);
        // the group's leader fetches the row id of the matching slot and the key image
        if (inGroup == 0) {
          const uint32_t slot = (mineMask ? static_cast<uint32_t>(h[u] >> 20) : 0u) & 15u;
          const uint32_t id = *reinterpret_cast<const uint32_t*>(buckets + (h[u] % numBuckets) * 128 + 16 + slot * 4);
          const uint4 img = images[mix(id ^ h[u]) % numKeys];
          t[u] = (img.x == static_cast<uint32_t>(h[u])) + img.y + img.z + img.w;
        }
        const uint32_t answer = __shfl(t[u], group * G, 64);
        if (ln == j + u * kPerWave + group) {
          result = answer;
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
    printf("%-34s table %8.1f MiB grid %5d: %7.3f ms  %7.1f G probes/s  %6.0f GB/s at 12 B/probe\n", name,
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
  if (argc > 1 && std::string(argv[1]) == "taggroup") {
    // generic (kHash) mode bucket shapes with key confirmation: lane per probe vs sub-wave groups
    for (uint64_t s : {64ULL << 20, 512ULL << 20, 4ULL << 30}) {
      run("slotkey  U4 (lane/probe)", (k_slotkey<4>), s, 2048);
      run("slotkey  U8 (lane/probe)", (k_slotkey<8>), s, 2048);
      run("tagswar  U4 (lane/probe)", (k_tagswar<4>), s, 2048);
      run("tagswar  U8 (lane/probe)", (k_tagswar<8>), s, 2048);
      run("taggroup G16 U2 (4 probes/wave)", (k_taggroup<16, 2>), s, 2048);
      run("taggroup G16 U4 (4 probes/wave)", (k_taggroup<16, 4>), s, 2048);
      run("taggroup G8  U2 (8 probes/wave)", (k_taggroup<8, 2>), s, 2048);
      run("taggroup G8  U4 (8 probes/wave)", (k_taggroup<8, 4>), s, 2048);
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
