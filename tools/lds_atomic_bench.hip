// Micro-benchmark behind DESIGN.md's accumulator strategy: cost per row of the
// candidate ways to fold one DOUBLE value per accumulator into a handful of
// groups inside one workgroup. Build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o /tmp/ldsb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int A = 6;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k(const double* in, const int* slots, double* out, long n, int S, int REP) {
  extern __shared__ double acc[];
  for (int i = threadIdx.x; i < S * A * REP; i += blockDim.x) acc[i] = 0;
  __syncthreads();
  const int rep = (threadIdx.x & 63) & (REP - 1);
  double r[4][A] = {};
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double v = in[i];
    const int s = slots[i];
    if (MODE == 0) {        // ds_add_f64, lane-replicated
      for (int j = 0; j < A; ++j) unsafeAtomicAdd(&acc[(s * A + j) * REP + rep], v + j);
    } else if (MODE == 1) { // ds_add_u64
      for (int j = 0; j < A; ++j) atomicAdd((unsigned long long*)&acc[(s * A + j) * REP + rep], (unsigned long long)i + j);
    } else if (MODE == 2) { // ds_add_f32 (2 per value would be needed)
      for (int j = 0; j < A; ++j) atomicAdd((float*)&acc[(s * A + j) * REP + rep], (float)v + j);
    } else if (MODE == 3) { // registers, predicated adds over 4 slots
      for (int g = 0; g < 4; ++g)
        for (int j = 0; j < A; ++j) r[g][j] += (s == g) ? v + j : 0.0;
    } else if (MODE == 4) { // loads only
      r[0][0] += v + s;
    } else if (MODE == 5) { // non-atomic LDS read-modify-write (racy across waves; cost probe only)
      for (int j = 0; j < A; ++j) { double* p = &acc[(s * A + j) * REP + rep]; *p = *p + v + j; }
    }
  }
  if (MODE == 3 || MODE == 4) {
    double t = 0;
    for (int g = 0; g < 4; ++g) for (int j = 0; j < A; ++j) t += r[g][j];
    if (t == 12345.678) out[0] = t;
  }
  __syncthreads();
  if (threadIdx.x < S * A) {
    double t = 0;
    for (int q = 0; q < REP; ++q) t += acc[threadIdx.x * REP + q];
    unsafeAtomicAdd(&out[threadIdx.x], t);
  }
}

int main() {
  const long n = 1L << 28;
  double* in; int* slots; double* out;
  CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&slots, n * 4)); CK(hipMalloc(&out, 4096 * 8));
  std::vector<int> hs(1 << 20);
  for (size_t i = 0; i < hs.size(); ++i) hs[i] = (i * 2654435761u >> 13) & 3;
  for (long off = 0; off < n; off += (long)hs.size()) CK(hipMemcpy(slots + off, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(in, 0, n * 8)); CK(hipMemset(out, 0, 4096 * 8));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](const char* name, auto kern, int S, int REP, int block, int grid) {
    size_t lds = (size_t)S * A * REP * 8;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, 0, in, slots, out, n, S, REP);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, 0, in, slots, out, n, S, REP);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-34s S=%d REP=%2d block=%4d grid=%4d: %7.3f ms  %6.1f Grows/s  %6.0f GB/s(12B/row)\n", name, S, REP, block, grid, ms,
           n / ms / 1e6, n * 12.0 / ms / 1e6);
  };
  for (int rep : {64, 16, 1}) {
    run("ds_add_f64", k<0>, 4, rep, 1024, 512);
    run("ds_add_u64", k<1>, 4, rep, 1024, 512);
    run("ds_add_f32", k<2>, 4, rep, 1024, 512);
    run("lds rmw non-atomic", k<5>, 4, rep, 1024, 512);
  }
  run("registers predicated", k<3>, 4, 1, 1024, 512);
  run("loads only", k<4>, 4, 1, 1024, 512);
  run("ds_add_f64 block256", k<0>, 4, 64, 256, 2048);
  run("registers predicated block256", k<3>, 4, 1, 256, 2048);
  return 0;
}
