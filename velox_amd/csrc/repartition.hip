// vx355_partition_scatter: the local half of a repartitioned exchange
// (exec/PartitionedOutput.cpp: rows grouped by HashPartitionFunction's
// partition number before they leave for their consumer). Two launches over
// tiles of 4096 rows: per-tile partition histograms, then — after an exclusive
// scan laid out partition-major, tile-minor, so every (tile, partition) pair
// owns a contiguous, ordered output range — a stable scatter in which each lane
// ranks its row among the rows of the same partition in its wave with ballots.
#include "common.h"

namespace vx {
namespace {

constexpr int kMaxParts = 64;
constexpr int kMaxScatterCols = 16;
constexpr int kTile = 4096;  // rows per workgroup (256 lanes x 16)

__global__ __launch_bounds__(256) void k_part_hist(const uint32_t* parts, int64_t numRows, int32_t numParts,
                                                   int64_t numTiles, uint32_t* tileCounts, uint32_t* badFlag) {
  __shared__ uint32_t hist[kMaxParts];
  for (int64_t tile = blockIdx.x; tile < numTiles; tile += gridDim.x) {
    if (threadIdx.x < kMaxParts) {
      hist[threadIdx.x] = 0;
    }
    blockSync();
    for (int j = 0; j < kTile / 256; ++j) {
      const int64_t r = tile * kTile + j * 256 + threadIdx.x;
      if (r < numRows) {
        const uint32_t p = parts[r];
        if (p < static_cast<uint32_t>(numParts)) {
          atomicAdd(&hist[p], 1u);
        } else {
          *badFlag = 1;  // reported as VX355_EINVAL; the scatter skips the row
        }
      }
    }
    blockSync();
    if (threadIdx.x < numParts) {
      // partition-major layout: counts[p * numTiles + tile]
      tileCounts[static_cast<int64_t>(threadIdx.x) * numTiles + tile] = hist[threadIdx.x];
    }
    blockSync();
  }
}

// Exclusive scan of numParts * numTiles counts by one workgroup.
__global__ __launch_bounds__(1024) void k_part_scan(const uint32_t* counts, int64_t n, uint64_t* offsets) {
  __shared__ uint64_t partial[1024];
  const int t = threadIdx.x;
  const int64_t per = (n + blockDim.x - 1) / blockDim.x;
  const int64_t begin = t * per;
  const int64_t end = begin + per < n ? begin + per : n;
  uint64_t sum = 0;
  for (int64_t i = begin; i < end; ++i) {
    sum += counts[i];
  }
  partial[t] = sum;
  blockSync();
  for (int off = 1; off < 1024; off <<= 1) {
    uint64_t v = t >= off ? partial[t - off] : 0;
    blockSync();
    partial[t] += v;
    blockSync();
  }
  uint64_t run = t == 0 ? 0 : partial[t - 1];
  for (int64_t i = begin; i < end; ++i) {
    offsets[i] = run;
    run += counts[i];
  }
  if (t == blockDim.x - 1) {
    offsets[n] = partial[1023];
  }
}

struct ScatterArgs {
  const uint32_t* parts;
  int64_t numRows;
  int32_t numParts;
  int32_t numCols;
  int64_t numTiles;
  const uint64_t* offsets;  // [p * numTiles + tile]
  const char* in[kMaxScatterCols];
  char* out[kMaxScatterCols];
  int32_t width[kMaxScatterCols];
};

__global__ __launch_bounds__(256) void k_part_scatter(ScatterArgs a) {
  __shared__ uint32_t running[kMaxParts];    // rows of the tile already placed, per partition
  __shared__ uint32_t waveCount[4][kMaxParts];
  for (int64_t tile = blockIdx.x; tile < a.numTiles; tile += gridDim.x) {
    if (threadIdx.x < kMaxParts) {
      running[threadIdx.x] = 0;
    }
    blockSync();
    const int wave = threadIdx.x >> 6;
    for (int j = 0; j < kTile / 256; ++j) {
      const int64_t r = tile * kTile + j * 256 + threadIdx.x;
      uint32_t p = r < a.numRows ? a.parts[r] : 0xffffffffu;
      const bool live = p < static_cast<uint32_t>(a.numParts);
      p = live ? p : 0xffffffffu;
      // Rank among the lanes of this wave with the same partition, and the
      // wave's count per partition.
      uint32_t rank = 0;
      for (int q = 0; q < a.numParts; ++q) {
        const uint64_t m = ballot(p == static_cast<uint32_t>(q));
        if (p == static_cast<uint32_t>(q)) {
          rank = static_cast<uint32_t>(lanePrefix(m));
        }
        if (lane() == 0) {
          waveCount[wave][q] = static_cast<uint32_t>(popc64(m));
        }
      }
      blockSync();
      if (live) {
        uint32_t before = running[p];
        for (int w = 0; w < wave; ++w) {
          before += waveCount[w][p];
        }
        const uint64_t dst = a.offsets[static_cast<int64_t>(p) * a.numTiles + tile] + before + rank;
        for (int c = 0; c < a.numCols; ++c) {
          const int w = a.width[c];
          const char* src = a.in[c] + r * w;
          char* d = a.out[c] + dst * w;
          if (w == 8) {
            *reinterpret_cast<uint64_t*>(d) = *reinterpret_cast<const uint64_t*>(src);
          } else if (w == 4) {
            *reinterpret_cast<uint32_t*>(d) = *reinterpret_cast<const uint32_t*>(src);
          } else if (w == 16) {
            *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(src);
          } else {
            for (int b = 0; b < w; ++b) {
              d[b] = src[b];
            }
          }
        }
      }
      blockSync();
      if (threadIdx.x < a.numParts) {
        running[threadIdx.x] += waveCount[0][threadIdx.x] + waveCount[1][threadIdx.x] +
            waveCount[2][threadIdx.x] + waveCount[3][threadIdx.x];
      }
      blockSync();
    }
  }
}

}  // namespace
}  // namespace vx

using namespace vx;

extern "C" int vx355_partition_scatter(const uint32_t* partitions, int32_t num_rows, int32_t num_partitions,
                                       const void* const* cols_in, const int32_t* widths, int32_t num_cols,
                                       void* const* cols_out, int64_t* counts_out, int32_t mem) {
  VX_API_BEGIN
  auto& rt = Runtime::get();
  rt.requireInit();
  VX_CHECK_ARG(num_rows >= 0, "negative num_rows");
  VX_CHECK_ARG(num_partitions >= 1 && num_partitions <= kMaxParts, "1..64 partitions");
  VX_CHECK_ARG(num_cols >= 0 && num_cols <= kMaxScatterCols, "at most 16 columns");
  VX_CHECK_ARG(counts_out != nullptr, "counts_out is NULL");
  for (int32_t p = 0; p < num_partitions; ++p) {
    counts_out[p] = 0;
  }
  if (num_rows == 0) {
    return VX355_OK;
  }
  VX_CHECK_ARG(partitions && (num_cols == 0 || (cols_in && cols_out && widths)), "NULL argument");
  const bool host = mem == VX355_MEM_HOST;
  const int64_t n = num_rows;
  const int64_t numTiles = ceilDiv(n, kTile);
  DevBuf dParts, dCounts, dOffsets;
  std::vector<DevBuf> dIn(num_cols), dOut(num_cols);
  const uint32_t* parts = partitions;
  if (host) {
    copyIn(dParts.ensure(static_cast<size_t>(n) * 4 + 64), partitions, VX355_MEM_HOST, static_cast<size_t>(n) * 4);
    parts = dParts.as<uint32_t>();
  }
  ScatterArgs sa{};
  sa.parts = parts;
  sa.numRows = n;
  sa.numParts = num_partitions;
  sa.numCols = num_cols;
  sa.numTiles = numTiles;
  for (int32_t c = 0; c < num_cols; ++c) {
    const int32_t w = widths[c];
    VX_CHECK_ARG(w == 1 || w == 2 || w == 4 || w == 8 || w == 16, "column width must be 1, 2, 4, 8 or 16");
    VX_CHECK_ARG(cols_in[c] && cols_out[c], "NULL column");
    sa.width[c] = w;
    if (host) {
      copyIn(dIn[c].ensure(static_cast<size_t>(n) * w + 64), cols_in[c], VX355_MEM_HOST, static_cast<size_t>(n) * w);
      sa.in[c] = dIn[c].as<char>();
      sa.out[c] = static_cast<char*>(dOut[c].ensure(static_cast<size_t>(n) * w + 64));
    } else {
      sa.in[c] = static_cast<const char*>(cols_in[c]);
      sa.out[c] = static_cast<char*>(cols_out[c]);
    }
  }
  const int64_t cells = numTiles * num_partitions;
  uint32_t* counts = static_cast<uint32_t*>(dCounts.ensure(static_cast<size_t>(cells + 1) * 4 + 64));
  uint32_t* badFlag = counts + cells;
  HIP_OK(hipMemsetAsync(badFlag, 0, 4, rt.stream));
  uint64_t* offsets = static_cast<uint64_t*>(dOffsets.ensure(static_cast<size_t>(cells + 1) * 8 + 64));
  const int grid = static_cast<int>(std::min<int64_t>(numTiles, static_cast<int64_t>(rt.numCUs) * 8));
  VX_LAUNCH("k_part_hist", k_part_hist, grid, 256, 0, parts, n, num_partitions, numTiles, counts, badFlag);
  VX_LAUNCH("k_part_scan", k_part_scan, 1, 1024, 0, counts, cells, offsets);
  sa.offsets = offsets;
  VX_LAUNCH("k_part_scatter", k_part_scatter, grid, 256, 0, sa);
  // Partition sizes = differences of the partition-major offsets.
  std::vector<uint64_t> firsts(num_partitions + 1);
  for (int32_t p = 0; p < num_partitions; ++p) {
    copyOutAsync(&firsts[p], VX355_MEM_HOST, offsets + static_cast<int64_t>(p) * numTiles, 8);
  }
  copyOutAsync(&firsts[num_partitions], VX355_MEM_HOST, offsets + cells, 8);
  uint32_t bad = 0;
  copyOutAsync(&bad, VX355_MEM_HOST, badFlag, 4);
  if (host) {
    for (int32_t c = 0; c < num_cols; ++c) {
      copyOutAsync(cols_out[c], VX355_MEM_HOST, sa.out[c], static_cast<size_t>(n) * sa.width[c]);
    }
  }
  rt.sync();
  if (bad) {
    VX_THROW(VX355_EINVAL, "partitions[] holds a value >= num_partitions");
  }
  for (int32_t p = 0; p < num_partitions; ++p) {
    counts_out[p] = static_cast<int64_t>(firsts[p + 1] - firsts[p]);
  }
  VX_API_END
}
