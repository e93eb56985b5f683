// vx355_presto_serialize: the rows PartitionedOutput routes to each destination, written as
// PrestoPages (serializers/PrestoSerializer.h, VectorStream::flush in
// serializers/VectorStream.cpp:207-299) by the GPU. The wire format keeps only the non-null
// values of a column, in row order, behind an MSB-first null bitmap, so a page is a compaction:
//   1. k_page_count: per (tile of 2048 page rows, column) the number of non-null rows and, for
//      strings, their bytes;
//   2. the host turns those counts into the byte layout of every page (sizes are data
//      dependent: hasNulls decides whether a bitmap exists at all);
//   3. k_page_write: every (tile, column) compacts its values to their final place - null
//      bytes, end offsets, values / string bytes - and k_page_patch drops in the few fixed
//      bytes (page header, encoding names, counts).
// The bytes never leave HBM before the page is complete.
#include <cstring>

#include "common.h"

namespace vx {
namespace {

constexpr int kPageTile = 2048;  // page rows per workgroup: 256 lanes x 8 rows = whole null bytes
constexpr int kRowsPerLane = 8;

struct PageTile {
  int64_t rowBegin;      // position in rows[] (or batch row) of the tile's first row
  int32_t count;         // rows in the tile
  int32_t pageRowBegin;  // index of the first row inside its page
};

struct TileOut {
  uint64_t nullPos;     // start of the page column's null bytes; ~0 = the column has no nulls
  uint64_t valuePos;    // where this tile's first non-null value (or string byte) goes
  uint64_t offsetsPos;  // VARIABLE_WIDTH: start of the page column's end-offset array
  uint64_t bytesBefore;  // VARIABLE_WIDTH: string bytes of the page column before this tile
};

struct PagePatch {
  uint64_t pos;
  uint32_t len;
  uint32_t pad;
  unsigned char data[32];
};

struct PageArgs {
  const ColView* cols;
  const int32_t* rows;  // may be null
  const PageTile* tiles;
  int64_t numTiles;
  int32_t numCols;
  int32_t lossless;
  uint64_t* counts;       // [col * numTiles + tile] * 2: non-null rows, string bytes
  const TileOut* layout;  // [col * numTiles + tile]
  unsigned char* out;
  uint32_t* errorFlag;
};

struct __attribute__((packed)) Packed16 {
  uint16_t v;
};
struct __attribute__((packed)) Packed32 {
  uint32_t v;
};
struct __attribute__((packed)) Packed64 {
  uint64_t v;
};

__device__ inline bool isStringKind(int32_t k) { return k == VX355_VARCHAR || k == VX355_VARBINARY; }

// Exclusive prefix over the 256 lanes of the workgroup; *total = sum of all.
__device__ inline uint64_t blockExclusive(uint64_t v, uint64_t* lds, uint64_t* total) {
  const int t = threadIdx.x;
  lds[t] = v;
  blockSync();
  for (int off = 1; off < 256; off <<= 1) {
    const uint64_t add = t >= off ? lds[t - off] : 0;
    blockSync();
    lds[t] += add;
    blockSync();
  }
  const uint64_t inclusive = lds[t];
  *total = lds[255];
  blockSync();
  return inclusive - v;
}

__global__ __launch_bounds__(256) void k_page_count(PageArgs a) {
  __shared__ uint64_t lds[8];
  const int64_t tile = blockIdx.x;
  const int col = blockIdx.y;
  const ColView c = a.cols[col];
  const PageTile pt = a.tiles[tile];
  const bool str = isStringKind(c.kind);
  uint64_t nonNull = 0, bytes = 0;
  for (int j = 0; j < kRowsPerLane; ++j) {
    // strings: lane-blocked like k_page_write's string branch; fixed width: lane-cyclic (coalesced)
    const int r = str ? threadIdx.x * kRowsPerLane + j : j * 256 + static_cast<int>(threadIdx.x);
    if (r < pt.count) {
      const int64_t pos = pt.rowBegin + r;
      const int64_t row = a.rows ? a.rows[pos] : pos;
      if (!colIsNull(c, row)) {
        ++nonNull;
        if (str) {
          bytes += loadView(c, colIndex(c, row)).size;
        }
      }
    }
  }
  // workgroup totals: wave shuffles, then four partial sums through LDS
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    nonNull += shfl64(nonNull, lane() ^ off);
    bytes += shfl64(bytes, lane() ^ off);
  }
  if (lane() == 0) {
    lds[(threadIdx.x >> 6) * 2] = nonNull;
    lds[(threadIdx.x >> 6) * 2 + 1] = bytes;
  }
  blockSync();
  if (threadIdx.x == 0) {
    uint64_t* o = a.counts + (static_cast<int64_t>(col) * a.numTiles + tile) * 2;
    o[0] = lds[0] + lds[2] + lds[4] + lds[6];
    o[1] = lds[1] + lds[3] + lds[5] + lds[7];
  }
}

__device__ inline const unsigned char* viewBytes(const StringView16& v, const uint4* slot) {
  // inline strings live in the view itself: prefix (4) + tail (8)
  return v.size <= 12 ? reinterpret_cast<const unsigned char*>(slot) + 4
                      : reinterpret_cast<const unsigned char*>(v.tail);
}

__global__ __launch_bounds__(256) void k_page_write(PageArgs a) {
  __shared__ uint64_t lds[256];
  const int64_t tile = blockIdx.x;
  const int col = blockIdx.y;
  const ColView c = a.cols[col];
  const PageTile pt = a.tiles[tile];
  const TileOut lo = a.layout[static_cast<int64_t>(col) * a.numTiles + tile];
  const bool str = isStringKind(c.kind);
  if (!str) {
    // Fixed width. A wave owns 512 consecutive rows of the tile and lane l takes rows
    // base + j * 64 + l: every load instruction reads 64 consecutive values, the validity of
    // 64 rows is one ballot (= 8 finished null bytes), and the compacted position of a value
    // is the count of valid rows before it: earlier waves, earlier ballots, lower lanes.
    __shared__ uint32_t waveTotals[4];
    const int wave = threadIdx.x >> 6;
    const int ln = lane();
    const int waveBase = wave * (kPageTile / 4);
    uint64_t ballots[kRowsPerLane];
    int64_t rowOf[kRowsPerLane];
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < kRowsPerLane; ++j) {
      const int r = waveBase + j * 64 + ln;
      bool v = false;
      rowOf[j] = -1;
      if (r < pt.count) {
        const int64_t pos = pt.rowBegin + r;
        rowOf[j] = a.rows ? a.rows[pos] : pos;
        v = !colIsNull(c, rowOf[j]);
      }
      ballots[j] = ballot(v);
      mine += static_cast<uint32_t>(popc64(ballots[j]));
      if (lo.nullPos != ~0ULL && ln < 8) {
        // wire polarity: 1 = null, first row in the most significant bit (ByteOutputStream with
        // isReverseBitOrder, VectorStream.cpp:64); bits past the last row stay 0
        const int firstRow = waveBase + j * 64 + ln * 8;
        if (firstRow < pt.count) {
          const uint32_t validBits = static_cast<uint32_t>(ballots[j] >> (ln * 8)) & 0xffu;
          const int rowsHere = pt.count - firstRow < 8 ? pt.count - firstRow : 8;
          uint32_t byte = 0;
          for (int k = 0; k < rowsHere; ++k) {
            if (!((validBits >> k) & 1)) {
              byte |= 0x80u >> k;
            }
          }
          a.out[lo.nullPos + static_cast<uint64_t>(pt.pageRowBegin + firstRow) / 8] = static_cast<unsigned char>(byte);
        }
      }
    }
    if (ln == 0) {
      waveTotals[wave] = mine;
    }
    blockSync();
    uint64_t run = 0;
    for (int w2 = 0; w2 < wave; ++w2) {
      run += waveTotals[w2];
    }
    const int w = c.kind == VX355_BOOLEAN || c.kind == VX355_TINYINT ? 1
        : c.kind == VX355_SMALLINT                                   ? 2
        : (c.kind == VX355_INTEGER || c.kind == VX355_REAL)          ? 4
        : c.kind == VX355_TIMESTAMP                                  ? (a.lossless ? 16 : 8)
                                                                     : 8;
#pragma unroll
    for (int j = 0; j < kRowsPerLane; ++j) {
      const uint64_t at = run + static_cast<uint64_t>(lanePrefix(ballots[j]));
      run += static_cast<uint64_t>(popc64(ballots[j]));
      if (!((ballots[j] >> ln) & 1)) {
        continue;
      }
      const int64_t i = colIndex(c, rowOf[j]);
      unsigned char* dst = a.out + lo.valuePos + at * w;
      switch (c.kind) {
        case VX355_BOOLEAN:
          *dst = bitAt(static_cast<const uint64_t*>(c.values), i) ? 1 : 0;
          break;
        case VX355_TINYINT:
          *dst = static_cast<const unsigned char*>(c.values)[i];
          break;
        case VX355_SMALLINT:
          reinterpret_cast<Packed16*>(dst)->v = static_cast<const uint16_t*>(c.values)[i];
          break;
        case VX355_INTEGER:
        case VX355_REAL:
          reinterpret_cast<Packed32*>(dst)->v = static_cast<const uint32_t*>(c.values)[i];
          break;
        case VX355_TIMESTAMP: {
          const int64_t seconds = static_cast<const int64_t*>(c.values)[i * 2];
          const uint64_t nanos = static_cast<const uint64_t*>(c.values)[i * 2 + 1];
          if (a.lossless) {
            reinterpret_cast<Packed64*>(dst)->v = static_cast<uint64_t>(seconds);
            reinterpret_cast<Packed64*>(dst + 8)->v = nanos;
          } else {
            // Timestamp::toMillis (type/Timestamp.h:157-172)
            const __int128_t ms = static_cast<__int128_t>(seconds) * 1000 + static_cast<int64_t>(nanos / 1000000);
            if (ms < INT64_MIN || ms > INT64_MAX) {
              *a.errorFlag = 1;
            }
            reinterpret_cast<Packed64*>(dst)->v = static_cast<uint64_t>(static_cast<int64_t>(ms));
          }
          break;
        }
        default:
          reinterpret_cast<Packed64*>(dst)->v = static_cast<const uint64_t*>(c.values)[i];
      }
    }
    return;
  }
  // VARIABLE_WIDTH: lane-blocked, 8 consecutive rows per lane (one null byte per lane)
  const int first = threadIdx.x * kRowsPerLane;
  uint32_t valid = 0;
  int64_t rowOf[kRowsPerLane];
  for (int j = 0; j < kRowsPerLane; ++j) {
    rowOf[j] = -1;
    const int r = first + j;
    if (r < pt.count) {
      const int64_t pos = pt.rowBegin + r;
      rowOf[j] = a.rows ? a.rows[pos] : pos;
      if (!colIsNull(c, rowOf[j])) {
        valid |= 1u << j;
      }
    }
  }
  if (lo.nullPos != ~0ULL && first < pt.count) {
    uint32_t byte = 0;
    for (int j = 0; j < kRowsPerLane; ++j) {
      if (first + j < pt.count && !((valid >> j) & 1)) {
        byte |= 0x80u >> j;
      }
    }
    a.out[lo.nullPos + static_cast<uint64_t>(pt.pageRowBegin + first) / 8] = static_cast<unsigned char>(byte);
  }
  // VARIABLE_WIDTH: end offset of every row (a null repeats the previous one), then the bytes
  uint64_t mine = 0;
  uint32_t sizes[kRowsPerLane];
  for (int j = 0; j < kRowsPerLane; ++j) {
    sizes[j] = 0;
    if ((valid >> j) & 1) {
      sizes[j] = loadView(c, colIndex(c, rowOf[j])).size;
      mine += sizes[j];
    }
  }
  uint64_t total;
  uint64_t before = blockExclusive(mine, lds, &total);
  uint64_t run = lo.bytesBefore + before;
  unsigned char* dst = a.out + lo.valuePos + before;
  for (int j = 0; j < kRowsPerLane; ++j) {
    const int r = first + j;
    if (r >= pt.count) {
      break;
    }
    if ((valid >> j) & 1) {
      const int64_t i = colIndex(c, rowOf[j]);
      const uint4* slot = static_cast<const uint4*>(c.values) + i;
      const StringView16 v = loadView(c, i);
      const unsigned char* src = viewBytes(v, slot);
      for (uint32_t b = 0; b < sizes[j]; ++b) {
        dst[b] = src[b];
      }
      dst += sizes[j];
      run += sizes[j];
    }
    reinterpret_cast<Packed32*>(a.out + lo.offsetsPos + 4ULL * static_cast<uint64_t>(pt.pageRowBegin + r))->v =
        static_cast<uint32_t>(run);
  }
}

__global__ __launch_bounds__(256) void k_page_patch(const PagePatch* patches, int64_t n, unsigned char* out) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) {
    return;
  }
  const PagePatch p = patches[i];
  for (uint32_t b = 0; b < p.len; ++b) {
    out[p.pos + b] = p.data[b];
  }
}

// folly::crc32 / boost crc_32_type as velox/common/base/Crc.h wraps it: reflected polynomial
// 0xEDB88320, initial value ~0, final complement.
struct Crc32 {
  uint32_t table[8][256];
  Crc32() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) {
        c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      }
      table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i) {
      for (int t = 1; t < 8; ++t) {
        table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xff];
      }
    }
  }
  uint32_t update(uint32_t state, const unsigned char* p, size_t n) const {
    while (n >= 8) {
      uint32_t lo, hi;
      std::memcpy(&lo, p, 4);
      std::memcpy(&hi, p + 4, 4);
      lo ^= state;
      state = table[7][lo & 0xff] ^ table[6][(lo >> 8) & 0xff] ^ table[5][(lo >> 16) & 0xff] ^ table[4][lo >> 24] ^
          table[3][hi & 0xff] ^ table[2][(hi >> 8) & 0xff] ^ table[1][(hi >> 16) & 0xff] ^ table[0][hi >> 24];
      p += 8;
      n -= 8;
    }
    while (n--) {
      state = table[0][(state ^ *p++) & 0xff] ^ (state >> 8);
    }
    return state;
  }
};

const char* encodingName(int32_t kind) {
  // typeToEncodingName (PrestoSerializerSerializationUtils.cpp:997-1040)
  switch (kind) {
    case VX355_BOOLEAN:
    case VX355_TINYINT:
      return "BYTE_ARRAY";
    case VX355_SMALLINT:
      return "SHORT_ARRAY";
    case VX355_INTEGER:
    case VX355_REAL:
      return "INT_ARRAY";
    case VX355_BIGINT:
    case VX355_DOUBLE:
    case VX355_TIMESTAMP:
      return "LONG_ARRAY";
    case VX355_VARCHAR:
    case VX355_VARBINARY:
      return "VARIABLE_WIDTH";
    default:
      return nullptr;
  }
}

int valueWidth(int32_t kind, bool lossless) {
  switch (kind) {
    case VX355_BOOLEAN:
    case VX355_TINYINT:
      return 1;
    case VX355_SMALLINT:
      return 2;
    case VX355_INTEGER:
    case VX355_REAL:
      return 4;
    case VX355_TIMESTAMP:
      return lossless ? 16 : 8;
    default:
      return 8;
  }
}

void putI32(unsigned char* p, int64_t v) {
  const int32_t x = static_cast<int32_t>(v);
  std::memcpy(p, &x, 4);
}

constexpr int kPageHeader = 4 + 1 + 4 + 4 + 8;  // PrestoSerializerSerializationUtils.h:37-45

void serializePages(const vx355_batch* batch, const int32_t* rows, int32_t rowsMem, const int64_t* offsets,
                    int32_t numPages, int32_t flags, void* out, int64_t outCapacity, int32_t outMem,
                    int64_t* pageOffsets) {
  auto& rt = Runtime::get();
  VX_CHECK_ARG(batch && offsets && pageOffsets && numPages >= 0, "NULL argument");
  const bool checksum = (flags & VX355_PAGE_CHECKSUM) != 0;
  const bool lossless = (flags & VX355_PAGE_LOSSLESS_TIMESTAMP) != 0;
  if (checksum && out && outMem != VX355_MEM_HOST) {
    VX_THROW(VX355_EUNSUPPORTED, "page checksums are computed for host output only");
  }
  const int32_t nc = batch->num_cols;
  VX_CHECK_ARG(nc >= 0 && (nc == 0 || batch->cols), "batch without columns array");
  std::vector<int32_t> used(nc);
  for (int32_t c = 0; c < nc; ++c) {
    used[c] = c;
    if (!encodingName(batch->cols[c].type_kind)) {
      VX_THROW(VX355_EUNSUPPORTED, "PrestoPage column of type kind " + std::to_string(batch->cols[c].type_kind));
    }
  }
  const int64_t limit = rows ? INT64_MAX : batch->num_rows;
  for (int32_t p = 0; p < numPages; ++p) {
    VX_CHECK_ARG(offsets[p] >= 0 && offsets[p + 1] >= offsets[p] && offsets[p + 1] <= limit, "page row ranges");
    VX_CHECK_ARG(offsets[p + 1] - offsets[p] <= INT32_MAX, "more than 2^31 rows in a page");
  }
  DeviceBatch db;
  db.load(batch, used);
  // tiles
  std::vector<PageTile> tiles;
  std::vector<int64_t> firstTile(numPages + 1, 0);
  for (int32_t p = 0; p < numPages; ++p) {
    firstTile[p] = static_cast<int64_t>(tiles.size());
    const int64_t n = offsets[p + 1] - offsets[p];
    for (int64_t b = 0; b < n; b += kPageTile) {
      tiles.push_back(PageTile{offsets[p] + b, static_cast<int32_t>(std::min<int64_t>(kPageTile, n - b)),
                               static_cast<int32_t>(b)});
    }
  }
  firstTile[numPages] = static_cast<int64_t>(tiles.size());
  const int64_t numTiles = static_cast<int64_t>(tiles.size());
  if (numTiles == 0) {
    for (int32_t p = 0; p <= numPages; ++p) {
      pageOffsets[p] = 0;
    }
    return;
  }
  DevBuf dRows, dTiles, dCols, dCounts, dLayout, dPatches, dOut, dFlag;
  const int32_t* devRows = nullptr;
  if (rows) {
    const int64_t last = offsets[numPages];
    if (rowsMem == VX355_MEM_HOST) {
      int32_t* staged = static_cast<int32_t*>(dRows.ensure(static_cast<size_t>(std::max<int64_t>(last, 1)) * 4 + 64));
      copyIn(staged, rows, VX355_MEM_HOST, static_cast<size_t>(last) * 4);
      devRows = staged;
    } else {
      devRows = rows;
    }
  }
  PageTile* devTiles = static_cast<PageTile*>(dTiles.ensure(tiles.size() * sizeof(PageTile) + 64));
  copyIn(devTiles, tiles.data(), VX355_MEM_HOST, tiles.size() * sizeof(PageTile));
  std::vector<ColView> views(std::max(nc, 1));
  for (int32_t c = 0; c < nc; ++c) {
    views[c] = db.col(c);
  }
  ColView* devCols = static_cast<ColView*>(dCols.ensure(views.size() * sizeof(ColView) + 64));
  copyIn(devCols, views.data(), VX355_MEM_HOST, views.size() * sizeof(ColView));
  const size_t numCells = static_cast<size_t>(numTiles) * std::max(nc, 1);
  uint64_t* devCounts = static_cast<uint64_t*>(dCounts.ensure(numCells * 16 + 64));
  uint32_t* devFlag = static_cast<uint32_t*>(dFlag.ensure(64));
  HIP_OK(hipMemsetAsync(devFlag, 0, 4, rt.stream));
  PageArgs a{};
  a.cols = devCols;
  a.rows = devRows;
  a.tiles = devTiles;
  a.numTiles = numTiles;
  a.numCols = nc;
  a.lossless = lossless ? 1 : 0;
  a.counts = devCounts;
  a.errorFlag = devFlag;
  std::vector<uint64_t> counts(numCells * 2, 0);
  if (nc > 0) {
    VX_LAUNCH("k_page_count", k_page_count, dim3(static_cast<unsigned>(numTiles), static_cast<unsigned>(nc)), 256, 0, a);
    copyOut(counts.data(), VX355_MEM_HOST, devCounts, numCells * 16);
  }
  // layout
  std::vector<TileOut> layout(numCells);
  std::vector<PagePatch> patches;
  auto patch = [&](uint64_t pos, const void* data, uint32_t len) {
    PagePatch pp{};
    pp.pos = pos;
    pp.len = len;
    std::memcpy(pp.data, data, len);
    patches.push_back(pp);
  };
  struct PageInfo {
    int64_t begin = 0, size = 0;
    int32_t rows = 0;
  };
  std::vector<PageInfo> pages(numPages);
  int64_t at = 0;
  for (int32_t p = 0; p < numPages; ++p) {
    const int64_t n = offsets[p + 1] - offsets[p];
    pages[p].begin = at;
    pages[p].rows = static_cast<int32_t>(n);
    pageOffsets[p] = at;
    if (n == 0) {
      continue;  // Destination::flush: nothing to send
    }
    int64_t pos = at + kPageHeader;
    unsigned char word[32];
    putI32(word, nc);
    patch(static_cast<uint64_t>(pos), word, 4);
    pos += 4;
    for (int32_t c = 0; c < nc; ++c) {
      const int32_t kind = batch->cols[c].type_kind;
      const char* name = encodingName(kind);
      const int32_t nameLen = static_cast<int32_t>(std::strlen(name));
      const bool str = isString(kind);
      uint64_t nonNull = 0, bytes = 0;
      for (int64_t t = firstTile[p]; t < firstTile[p + 1]; ++t) {
        const uint64_t* cell = &counts[(static_cast<size_t>(c) * numTiles + t) * 2];
        nonNull += cell[0];
        bytes += cell[1];
      }
      if (bytes > INT32_MAX) {
        VX_THROW(VX355_EUSER, "more than 2 GB of string bytes in one page column");
      }
      const bool hasNulls = nonNull < static_cast<uint64_t>(n);
      // header: name, row count (VectorStream::flush default / VARCHAR branches)
      putI32(word, nameLen);
      std::memcpy(word + 4, name, nameLen);
      putI32(word + 4 + nameLen, n);
      patch(static_cast<uint64_t>(pos), word, 8 + nameLen);
      pos += 8 + nameLen;
      uint64_t offsetsPos = 0;
      if (str) {
        offsetsPos = static_cast<uint64_t>(pos);
        pos += 4 * n;
      }
      const unsigned char flag = hasNulls ? 1 : 0;
      patch(static_cast<uint64_t>(pos), &flag, 1);
      pos += 1;
      uint64_t nullPos = ~0ULL;
      if (hasNulls) {
        nullPos = static_cast<uint64_t>(pos);
        pos += (n + 7) / 8;
      }
      if (str) {
        putI32(word, static_cast<int64_t>(bytes));
        patch(static_cast<uint64_t>(pos), word, 4);
        pos += 4;
      }
      const int w = str ? 1 : valueWidth(kind, lossless);
      uint64_t valueRun = 0, byteRun = 0;
      for (int64_t t = firstTile[p]; t < firstTile[p + 1]; ++t) {
        const size_t cellIndex = static_cast<size_t>(c) * numTiles + t;
        TileOut& lo = layout[cellIndex];
        lo.nullPos = nullPos;
        lo.offsetsPos = offsetsPos;
        lo.bytesBefore = byteRun;
        lo.valuePos = static_cast<uint64_t>(pos) + (str ? byteRun : valueRun * w);
        valueRun += counts[cellIndex * 2];
        byteRun += counts[cellIndex * 2 + 1];
      }
      pos += str ? static_cast<int64_t>(bytes) : static_cast<int64_t>(nonNull) * w;
    }
    pages[p].size = pos - at;
    if (pages[p].size - kPageHeader > INT32_MAX) {
      VX_THROW(VX355_EUSER, "page larger than 2 GB");
    }
    // page header; the checksum is filled in below
    unsigned char head[kPageHeader] = {0};
    putI32(head, n);
    head[4] = checksum ? 4 : 0;  // kCheckSumBitMask
    putI32(head + 5, pages[p].size - kPageHeader);
    putI32(head + 9, pages[p].size - kPageHeader);
    patch(static_cast<uint64_t>(at), head, kPageHeader);
    at = pos;
  }
  pageOffsets[numPages] = at;
  if (!out) {
    return;
  }
  VX_CHECK_ARG(outCapacity >= at, "output buffer smaller than the pages (call with out = NULL for the sizes)");
  if (at == 0) {
    return;
  }
  unsigned char* devOut = outMem == VX355_MEM_DEVICE ? static_cast<unsigned char*>(out)
                                                     : static_cast<unsigned char*>(dOut.ensure(static_cast<size_t>(at) + 64));
  a.out = devOut;
  if (nc > 0) {
    TileOut* devLayout = static_cast<TileOut*>(dLayout.ensure(layout.size() * sizeof(TileOut) + 64));
    copyIn(devLayout, layout.data(), VX355_MEM_HOST, layout.size() * sizeof(TileOut));
    a.layout = devLayout;
    VX_LAUNCH("k_page_write", k_page_write, dim3(static_cast<unsigned>(numTiles), static_cast<unsigned>(nc)), 256, 0, a);
  }
  PagePatch* devPatches = static_cast<PagePatch*>(dPatches.ensure(patches.size() * sizeof(PagePatch) + 64));
  copyIn(devPatches, patches.data(), VX355_MEM_HOST, patches.size() * sizeof(PagePatch));
  VX_LAUNCH("k_page_patch", k_page_patch, static_cast<int>(ceilDiv(static_cast<int64_t>(patches.size()), 256)), 256, 0,
            devPatches, static_cast<int64_t>(patches.size()), devOut);
  uint32_t bad = 0;
  copyOut(&bad, VX355_MEM_HOST, devFlag, 4);
  if (bad) {
    VX_THROW(VX355_EUSER, "Could not convert Timestamp to milliseconds");  // Timestamp::toMillis
  }
  if (outMem == VX355_MEM_HOST) {
    copyOut(out, VX355_MEM_HOST, devOut, static_cast<size_t>(at));
    if (checksum) {
      // computeChecksum (PrestoSerializerSerializationUtils.h:167-177): the listener sees the
      // bytes after the header, then codec, numRows, uncompressedSize
      static const Crc32 crc;
      unsigned char* base = static_cast<unsigned char*>(out);
      for (int32_t p = 0; p < numPages; ++p) {
        if (pages[p].rows == 0) {
          continue;
        }
        unsigned char* page = base + pages[p].begin;
        uint32_t state = ~0u;
        state = crc.update(state, page + kPageHeader, static_cast<size_t>(pages[p].size - kPageHeader));
        state = crc.update(state, page + 4, 1);
        state = crc.update(state, page, 4);
        state = crc.update(state, page + 5, 4);
        const int64_t sum = static_cast<int64_t>(static_cast<uint32_t>(~state));
        std::memcpy(page + 13, &sum, 8);
      }
    }
  } else {
    rt.sync();
  }
}

}  // namespace
}  // namespace vx

extern "C" {

int vx355_presto_serialize(const vx355_batch* batch, const int32_t* rows, int32_t rows_mem, const int64_t* offsets,
                           int32_t num_pages, int32_t flags, void* out, int64_t out_capacity, int32_t out_mem,
                           int64_t* page_offsets) {
  VX_API_BEGIN
  vx::Runtime::get().requireInit();
  vx::serializePages(batch, rows, rows_mem, offsets, num_pages, flags, out, out_capacity, out_mem, page_offsets);
  VX_API_END
}

}  // extern "C"
