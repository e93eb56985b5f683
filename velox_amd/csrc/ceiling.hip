// What the HBM of THIS box delivers to plain streaming kernels: the practical ceiling next to
// the 8 TB/s datasheet figure every roofline fraction of bench.py is quoted against. Two
// hand-written kernels in the style of the library's streaming passes (16-byte accesses, several
// independent accesses per lane in flight, a few workgroups per CU):
//   VX355_CEILING_READ  a read-only stream - nontemporal loads folded into one word per lane, one
//                       store per workgroup: the shape of k_agg_fast (68 B read per row, nothing
//                       written), bounded by the read rate alone;
//   VX355_CEILING_COPY  read + write of the same number of bytes: the shape of the radix scatters,
//                       the partitioned probe's record pass and the page writer.
// MI355X_MICROARCH.md measures ~6.3 TB/s for a float4 copy; torch's Tensor.copy_ (what bench.py
// used before) reaches 4.7-5.0 TB/s and therefore sat BELOW kernels it was meant to bound; so did this
// file's own first copy kernel (4.5-5.0): see k_ceiling_copy.
#include "common.h"
#include "device_utils.h"

namespace vx {
namespace {

typedef unsigned int U32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_ceiling_read(const U32x4* src, int64_t n, uint32_t* sink) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x * 4;
  U32x4 acc = {0u, 0u, 0u, 0u};
  int64_t i = (static_cast<int64_t>(blockIdx.x) * blockDim.x) * 4 + threadIdx.x;
  for (; i + 3 * static_cast<int64_t>(blockDim.x) < n; i += stride) {
    const U32x4 a = __builtin_nontemporal_load(src + i);
    const U32x4 b = __builtin_nontemporal_load(src + i + blockDim.x);
    const U32x4 c = __builtin_nontemporal_load(src + i + 2 * static_cast<int64_t>(blockDim.x));
    const U32x4 d = __builtin_nontemporal_load(src + i + 3 * static_cast<int64_t>(blockDim.x));
    acc ^= a ^ b ^ c ^ d;
  }
  for (; i < n; i += blockDim.x) {
    acc ^= __builtin_nontemporal_load(src + i);
  }
  const uint32_t v = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (v == 0x9e3779b9u) {  // (never true for the fill pattern: keeps the loads alive without a store per lane)
    sink[blockIdx.x] = v;
  }
}

// VX355_CEILING_READ_COLUMNS: the scan of TPC-H Q1 with nothing behind it - the seven column streams of
// k_agg_fast's plan (two 16-byte StringView columns of which the first 8 bytes of every view are loaded,
// one 4-byte date, four 8-byte doubles = 68 bytes of HBM traffic per row) in k_agg_fast's own mapping of
// rows to lanes (512 lanes, four rows per lane 512 rows apart, every load of an iteration issued before
// the first use, nontemporal, three workgroups per CU, grid-stride), the loaded words xor-ed together.
// Seven interleaved streams deliver less than one (round 6, tools/q1_stream_bench.hip: 6.3-6.5 TB/s on a
// box whose single stream reaches 6.8; 16-byte loads over four ADJACENT rows per lane only 5.1): this is
// the number k_agg_fast's arithmetic competes with, measured on the same box in the same process.
struct ColumnStreams {
  const uint64_t* view[2];
  const uint32_t* date;
  const uint64_t* f64[4];
};
constexpr int kColumnsUnroll = 4;
__global__ __launch_bounds__(512, 4) void k_ceiling_columns(ColumnStreams c, int64_t n, uint32_t* sink) {
  const int64_t tile = 512LL * kColumnsUnroll;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * tile;
  const int64_t rounds = (n + stride - 1) / stride;
  int64_t base = static_cast<int64_t>(blockIdx.x) * tile + threadIdx.x;
  uint64_t acc = 0;
  for (int64_t r = 0; r < rounds; ++r, base += stride) {
    int64_t row[kColumnsUnroll];
    uint64_t k0[kColumnsUnroll], k1[kColumnsUnroll], x[4][kColumnsUnroll];
    uint32_t d[kColumnsUnroll];
#pragma unroll
    for (int u = 0; u < kColumnsUnroll; ++u) {
      const int64_t t = base + u * 512LL;
      row[u] = t < n ? t : n - 1;  // clamped, as k_agg_fast does: no load under a per-lane predicate
    }
#pragma unroll
    for (int u = 0; u < kColumnsUnroll; ++u) {
      k0[u] = __builtin_nontemporal_load(c.view[0] + row[u] * 2);
    }
#pragma unroll
    for (int u = 0; u < kColumnsUnroll; ++u) {
      k1[u] = __builtin_nontemporal_load(c.view[1] + row[u] * 2);
    }
#pragma unroll
    for (int u = 0; u < kColumnsUnroll; ++u) {
      d[u] = __builtin_nontemporal_load(c.date + row[u]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int u = 0; u < kColumnsUnroll; ++u) {
        x[j][u] = __builtin_nontemporal_load(c.f64[j] + row[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < kColumnsUnroll; ++u) {
      acc ^= k0[u] ^ k1[u] ^ d[u] ^ x[0][u] ^ x[1][u] ^ x[2][u] ^ x[3][u];
    }
  }
  if (acc == 0x9e3779b97f4a7c15ULL) {  // (never true for the fill pattern)
    sink[blockIdx.x] = static_cast<uint32_t>(acc);
  }
}

// The copy shape that got closest to the guide's 6.3 TB/s on this chip (tools/copy_bench.hip,
// profiles/r05_copy_bench.txt: 5.8-5.9 TB/s at 4 GiB + 4 GiB, 6.1 at 1 + 1; the r04 kernel - 512 lanes, 8
// workgroups per CU, grid-stride, plain loads and stores - reached 4.5-5.0): 1024-lane workgroups, four
// per CU, each owning ONE contiguous range rounded to 2 MiB, eight 16-byte nontemporal loads in flight
// per lane, nontemporal stores.
constexpr int kCopyUnroll = 8;
__global__ __launch_bounds__(1024) void k_ceiling_copy(const U32x4* __restrict__ src, U32x4* __restrict__ dst, int64_t n) {
  const int64_t B = blockDim.x;
  int64_t per = (n + gridDim.x - 1) / gridDim.x;
  per = (per + 131071) / 131072 * 131072;   // 2 MiB of 16-byte elements
  int64_t i = blockIdx.x * per + threadIdx.x;
  const int64_t end = (blockIdx.x + 1) * per < n ? (blockIdx.x + 1) * per : n;
  for (; i + (kCopyUnroll - 1) * B < end; i += B * kCopyUnroll) {
    U32x4 v[kCopyUnroll];
#pragma unroll
    for (int u = 0; u < kCopyUnroll; ++u) {
      v[u] = __builtin_nontemporal_load(src + i + u * B);
    }
#pragma unroll
    for (int u = 0; u < kCopyUnroll; ++u) {
      __builtin_nontemporal_store(v[u], dst + i + u * B);
    }
  }
  for (; i < end; i += B) {
    dst[i] = src[i];
  }
}

}  // namespace
}  // namespace vx

using namespace vx;

extern "C" int vx355_hbm_ceiling(int32_t kind, size_t bytes, int32_t iterations, double* gbytes_per_second) {
  VX_API_BEGIN
  VX_CHECK_ARG(gbytes_per_second && iterations >= 1 && bytes >= (1u << 20), "bad argument");
  VX_CHECK_ARG(kind == VX355_CEILING_READ || kind == VX355_CEILING_COPY || kind == VX355_CEILING_READ_COLUMNS,
               "unknown ceiling kernel");
  auto& rt = Runtime::get();
  rt.requireInit();
  // (READ_COLUMNS: 'bytes' = all seven columns together, 68 bytes per row, rows a multiple of 64)
  const int64_t rows = kind == VX355_CEILING_READ_COLUMNS ? static_cast<int64_t>(bytes / 68) & ~63LL : 0;
  const int64_t n = kind == VX355_CEILING_READ_COLUMNS ? (rows * 68 + 15) / 16 : static_cast<int64_t>(bytes / 16);
  DevBuf src, dst, sink;
  src.ensure(static_cast<size_t>(n) * 16);
  HIP_OK(hipMemsetAsync(src.ptr(), 0x5a, static_cast<size_t>(n) * 16, rt.stream));
  ColumnStreams cols{};
  if (kind == VX355_CEILING_READ_COLUMNS) {
    const unsigned char* p = src.as<unsigned char>();
    cols.view[0] = reinterpret_cast<const uint64_t*>(p);
    cols.view[1] = reinterpret_cast<const uint64_t*>(p + rows * 16);
    for (int j = 0; j < 4; ++j) {
      cols.f64[j] = reinterpret_cast<const uint64_t*>(p + rows * (32 + 8 * j));
    }
    cols.date = reinterpret_cast<const uint32_t*>(p + rows * 64);
  }
  const int grid = rt.numCUs * 8;
  sink.ensure(static_cast<size_t>(grid) * 4 + 64);
  if (kind == VX355_CEILING_COPY) {
    dst.ensure(static_cast<size_t>(n) * 16);
  }
  hipEvent_t begin = rt.newEvent(), end = rt.newEvent();
  auto launch = [&]() {
    if (kind == VX355_CEILING_READ_COLUMNS) {
      hipLaunchKernelGGL(k_ceiling_columns, dim3(rt.numCUs * 3), dim3(512), 0, rt.stream, cols, rows, sink.as<uint32_t>());
    } else if (kind == VX355_CEILING_READ) {
      hipLaunchKernelGGL(k_ceiling_read, dim3(grid), dim3(512), 0, rt.stream, src.as<U32x4>(), n, sink.as<uint32_t>());
    } else {
      hipLaunchKernelGGL(k_ceiling_copy, dim3(rt.numCUs * 4), dim3(1024), 0, rt.stream, src.as<U32x4>(), dst.as<U32x4>(), n);
    }
  };
  launch();
  launch();  // warm-up: clocks, TLB
  HIP_OK(hipEventRecord(begin, rt.stream));
  for (int32_t i = 0; i < iterations; ++i) {
    launch();
  }
  HIP_OK(hipEventRecord(end, rt.stream));
  HIP_OK(hipEventSynchronize(end));
  HIP_OK(hipGetLastError());
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, begin, end));
  (void)hipEventDestroy(begin);
  (void)hipEventDestroy(end);
  const double moved = kind == VX355_CEILING_READ_COLUMNS
      ? static_cast<double>(rows) * 68 * iterations
      : static_cast<double>(n) * 16 * (kind == VX355_CEILING_COPY ? 2 : 1) * iterations;
  *gbytes_per_second = moved / (static_cast<double>(ms) * 1e-3) / 1e9;
  VX_API_END
}
