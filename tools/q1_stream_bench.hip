// What is the read rate of TPC-H Q1's seven column streams with NO arithmetic behind them?
// Same access pattern as k_agg_fast on Q1 (two 16-byte StringView columns of which the first 8 bytes of
// every view are loaded, one 4-byte date, four 8-byte doubles; nontemporal loads; 512 threads, UNROLL rows
// per lane, grid-stride), the loaded words xor-ed together. The ceiling k_agg_fast is measured against.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/q1_stream_bench tools/q1_stream_bench.hip && /tmp/q1_stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Cols {
  const uint64_t* k0; const uint64_t* k1; const uint32_t* d; const uint64_t* x[4];
};

// The same loads with every workgroup walking ONE contiguous range of rows (rows / grid each) instead of
// tiles a whole grid apart.
template <int UNROLL>
__global__ __launch_bounds__(512, 4) void k_streams_chunked(Cols c, int64_t n, uint64_t* out) {
  const int64_t tile = 512LL * UNROLL;
  const int64_t tiles = (n + tile - 1) / tile;
  const int64_t per = (tiles + gridDim.x - 1) / gridDim.x;
  const int64_t first = blockIdx.x * per;
  const int64_t last = first + per < tiles ? first + per : tiles;
  uint64_t acc = 0;
  for (int64_t t = first; t < last; ++t) {
    const int64_t base = t * tile + threadIdx.x;
    int64_t row[UNROLL];
    uint64_t a[UNROLL], b[UNROLL], x[UNROLL][4];
    uint32_t d[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { int64_t r = base + u * 512LL; row[u] = r < n ? r : n - 1; }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) a[u] = __builtin_nontemporal_load(c.k0 + row[u] * 2);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) b[u] = __builtin_nontemporal_load(c.k1 + row[u] * 2);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) d[u] = __builtin_nontemporal_load(c.d + row[u]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) x[u][j] = __builtin_nontemporal_load(c.x[j] + row[u]);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= a[u] ^ b[u] ^ d[u] ^ x[u][0] ^ x[u][1] ^ x[u][2] ^ x[u][3];
  }
  if (acc == 0x1234567887654321ULL) out[0] = acc;
}

template <int UNROLL, bool PIPE>
__global__ __launch_bounds__(512, 4) void k_streams(Cols c, int64_t n, uint64_t* out) {
  const int64_t tile = 512LL * UNROLL;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * tile;
  const int64_t rounds = (n + stride - 1) / stride;
  int64_t base = static_cast<int64_t>(blockIdx.x) * tile + threadIdx.x;
  uint64_t acc = 0;
  struct R { uint64_t a[UNROLL], b[UNROLL], x[UNROLL][4]; uint32_t d[UNROLL]; };
  auto load = [&](R& r, int64_t base) {
    int64_t row[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { int64_t t = base + u * 512LL; row[u] = t < n ? t : n - 1; }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) r.a[u] = __builtin_nontemporal_load(c.k0 + row[u] * 2);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) r.b[u] = __builtin_nontemporal_load(c.k1 + row[u] * 2);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) r.d[u] = __builtin_nontemporal_load(c.d + row[u]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) r.x[u][j] = __builtin_nontemporal_load(c.x[j] + row[u]);
  };
  auto use = [&](const R& r) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= r.a[u] ^ r.b[u] ^ r.d[u] ^ r.x[u][0] ^ r.x[u][1] ^ r.x[u][2] ^ r.x[u][3];
  };
  if (PIPE) {
    R ra, rb;
    load(ra, base);
    for (int64_t r = 0; r < rounds; r += 2, base += 2 * stride) {
      load(rb, base + stride); use(ra); load(ra, base + 2 * stride); use(rb);
    }
  } else {
    for (int64_t r = 0; r < rounds; ++r, base += stride) { R x; load(x, base); use(x); }
  }
  if (acc == 0x1234567887654321ULL) out[0] = acc;
}

// The same bytes with FOUR ADJACENT rows per lane: 16-byte loads (two doubles / one view / four dates per
// instruction), a wave covers 2 - 4 KB of a column per instruction instead of 512 B.
struct alignas(16) U4 { uint32_t a, b, c, d; };
__device__ inline U4 ntload16(const void* p) {
  typedef uint32_t v4 __attribute__((ext_vector_type(4)));
  v4 v = __builtin_nontemporal_load(static_cast<const v4*>(p));
  return U4{v.x, v.y, v.z, v.w};
}
template <bool FULLVIEW>
__global__ __launch_bounds__(512, 4) void k_streams_wide(Cols c, int64_t n, uint64_t* out) {
  const int64_t tile = 512LL * 4;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * tile;
  const int64_t rounds = (n + stride - 1) / stride;   // n is a multiple of 4 here
  int64_t base = static_cast<int64_t>(blockIdx.x) * tile + threadIdx.x * 4LL;
  uint32_t acc = 0;
  for (int64_t r = 0; r < rounds; ++r, base += stride) {
    const int64_t row = base + 3 < n ? base : n - 4;
    U4 k0[4], k1[4], d, x[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) k0[u] = ntload16(c.k0 + (row + u) * 2);
#pragma unroll
    for (int u = 0; u < 4; ++u) k1[u] = ntload16(c.k1 + (row + u) * 2);
    d = ntload16(c.d + row);
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[j][0] = ntload16(c.x[j] + row); x[j][1] = ntload16(c.x[j] + row + 2); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc ^= k0[u].a ^ k0[u].b ^ k1[u].a ^ k1[u].b;
      if (FULLVIEW) acc ^= k0[u].c ^ k0[u].d ^ k1[u].c ^ k1[u].d;
    }
    acc ^= d.a ^ d.b ^ d.c ^ d.d;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc ^= x[j][0].a ^ x[j][0].b ^ x[j][0].c ^ x[j][0].d ^ x[j][1].a ^ x[j][1].b ^ x[j][1].c ^ x[j][1].d;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const int64_t n = (argc > 1 ? atoll(argv[1]) : 300000000LL) & ~3LL;
  Cols c;
  void* p;
  OK(hipMalloc(&p, n * 16)); OK(hipMemset(p, 1, n * 16)); c.k0 = (const uint64_t*)p;
  OK(hipMalloc(&p, n * 16)); OK(hipMemset(p, 2, n * 16)); c.k1 = (const uint64_t*)p;
  OK(hipMalloc(&p, n * 4)); OK(hipMemset(p, 3, n * 4)); c.d = (const uint32_t*)p;
  for (int j = 0; j < 4; ++j) { OK(hipMalloc(&p, n * 8)); OK(hipMemset(p, 4 + j, n * 8)); c.x[j] = (const uint64_t*)p; }
  uint64_t* out; OK(hipMalloc(&out, 8));
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  const double bytes = 68.0 * n;
  auto run = [&](const char* name, auto kernel, int grid) {
    for (int i = 0; i < 3; ++i) kernel<<<grid, 512>>>(c, n, out);
    OK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) kernel<<<grid, 512>>>(c, n, out);
    OK(hipEventRecord(e1)); OK(hipEventSynchronize(e1));
    float ms; OK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s grid %5d  %.3f ms  %.0f GB/s\n", name, grid, ms / reps, bytes / (ms / reps) / 1e6);
  };
  for (int grid : {256, 384, 512, 640, 768, 1024, 1536}) {
    run("unroll 4", k_streams<4, false>, grid);
    run("unroll 2", k_streams<2, false>, grid);
    run("unroll 2 pipelined", k_streams<2, true>, grid);
    run("unroll 4 pipelined", k_streams<4, true>, grid);
    run("4 adjacent rows, 16-B loads", k_streams_wide<true>, grid);
    run("unroll 4, contiguous range", k_streams_chunked<4>, grid);
  }
  return 0;
}
