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

// The rows a column stream serialises and how they are cut into tiles. Top-level columns share the
// page rows; the children of a ROW column with a null bitmap see the rows of the NON-NULL structs
// only (serializeRowVector, PrestoSerializerSerializationUtils.cpp:883-919): their own row list,
// their own tiles.
struct ColTiles {
  const int32_t* rows;      // positions -> batch rows; null = the position is the row
  const PageTile* tiles;
  int64_t numTiles;
  int64_t cellBase;         // first cell of this stream in counts / layout (one cell per tile)
};

// ColView::kind of a ROW node: the stream carries the struct's null bits and, per row, the running
// number of non-null structs (the "offsets" of the ROW encoding) - a VARIABLE_WIDTH column whose
// every non-null value is one "byte" long and has no bytes.
constexpr int32_t kRowNodeKind = 1000;

struct PageArgs {
  const ColView* cols;        // every stream of the batch (top-level columns, ROW nodes, ROW children)
  const ColTiles* colTiles;   // per stream
  const int32_t* launchCols;  // blockIdx.y -> stream
  int32_t lossless;
  int32_t pad;
  uint64_t* counts;       // [cell] * 2: non-null rows, string bytes
  const TileOut* layout;  // [cell]
  unsigned char* out;
  uint32_t* errorFlag;    // 1: a timestamp does not fit milliseconds; 2: rows[] holds a row outside the batch
  int64_t batchRows;
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

// (lane-blocked streams: real strings and ROW nodes)
__device__ inline bool isStringKind(int32_t k) { return k == VX355_VARCHAR || k == VX355_VARBINARY || k == kRowNodeKind; }

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
  const int col = a.launchCols[blockIdx.y];
  const ColTiles ct = a.colTiles[col];
  if (tile >= ct.numTiles) {
    return;
  }
  const ColView c = a.cols[col];
  const PageTile pt = ct.tiles[tile];
  const bool str = isStringKind(c.kind);
  uint64_t nonNull = 0, bytes = 0;
  for (int j = 0; j < kRowsPerLane; ++j) {
    // strings: lane-blocked like k_page_write's string branch; fixed width: lane-cyclic (coalesced)
    const int r = str ? threadIdx.x * kRowsPerLane + j : j * 256 + static_cast<int>(threadIdx.x);
    if (r < pt.count) {
      const int64_t pos = pt.rowBegin + r;
      const int64_t row = ct.rows ? ct.rows[pos] : pos;
      if (row < 0 || row >= a.batchRows) {
        *a.errorFlag = 2;  // the host refuses the call before anything is written
        continue;
      }
      if (!colIsNull(c, row)) {
        ++nonNull;
        if (c.kind == kRowNodeKind) {
          bytes += 1;
        } else if (str) {
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
    uint64_t* o = a.counts + (ct.cellBase + tile) * 2;
    o[0] = lds[0] + lds[2] + lds[4] + lds[6];
    o[1] = lds[1] + lds[3] + lds[5] + lds[7];
  }
}

// The row list of a ROW column's children: the rows of the node's tile whose struct is not null,
// in order, at tileBase[tile] of 'out' (tileBase = non-null structs in the tiles before this one).
__global__ __launch_bounds__(256) void k_row_compact(ColView node, ColTiles ct, const uint64_t* tileBase, int32_t* out) {
  __shared__ uint64_t lds[256];
  const int64_t tile = blockIdx.x;
  const PageTile pt = ct.tiles[tile];
  const int first = threadIdx.x * kRowsPerLane;
  int32_t rowOf[kRowsPerLane];
  uint32_t valid = 0;
  for (int j = 0; j < kRowsPerLane; ++j) {
    const int r = first + j;
    rowOf[j] = -1;
    if (r < pt.count) {
      const int64_t pos = pt.rowBegin + r;
      rowOf[j] = static_cast<int32_t>(ct.rows ? ct.rows[pos] : pos);
      if (!colIsNull(node, rowOf[j])) {
        valid |= 1u << j;
      }
    }
  }
  uint64_t total;
  uint64_t at = tileBase[tile] + blockExclusive(static_cast<uint64_t>(__popc(valid)), lds, &total);
  for (int j = 0; j < kRowsPerLane; ++j) {
    if ((valid >> j) & 1) {
      out[at++] = rowOf[j];
    }
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
  const int col = a.launchCols[blockIdx.y];
  const ColTiles ct = a.colTiles[col];
  if (tile >= ct.numTiles) {
    return;
  }
  const ColView c = a.cols[col];
  const PageTile pt = ct.tiles[tile];
  const TileOut lo = a.layout[ct.cellBase + tile];
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
        rowOf[j] = ct.rows ? ct.rows[pos] : pos;
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
      rowOf[j] = ct.rows ? ct.rows[pos] : pos;
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
      sizes[j] = c.kind == kRowNodeKind ? 1u : loadView(c, colIndex(c, rowOf[j])).size;
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
    if (((valid >> j) & 1) && c.kind == kRowNodeKind) {
      run += 1;  // a non-null struct: one more row in every child stream, no bytes of its own
    } else if ((valid >> j) & 1) {
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

// One column stream of the pages: a top-level column, a ROW node or a child of a ROW node.
struct StreamNode {
  int32_t kind = 0;             // vx355_type_kind (VX355_ROW for a node)
  ColView view{};               // device view (ROW node: kRowNodeKind, nulls = the struct's bitmap)
  int group = 0;                // whose rows it serialises: 0 = the page rows
  std::vector<int> children;    // streams of a ROW node's fields
  int64_t cellBase = 0;
};

// The rows of one group of streams, page by page.
struct RowGroup {
  const int32_t* devRows = nullptr;   // null: position = batch row
  std::vector<int64_t> pageBegin;     // numPages + 1 positions into devRows
  std::vector<PageTile> tiles;
  std::vector<int64_t> firstTile;     // numPages + 1
  DevBuf rowsBuf, tilesBuf;
  const PageTile* devTiles = nullptr;

  void cut(int32_t numPages) {
    tiles.clear();
    firstTile.assign(numPages + 1, 0);
    for (int32_t p = 0; p < numPages; ++p) {
      firstTile[p] = static_cast<int64_t>(tiles.size());
      const int64_t n = pageBegin[p + 1] - pageBegin[p];
      for (int64_t b = 0; b < n; b += kPageTile) {
        tiles.push_back(PageTile{pageBegin[p] + b, static_cast<int32_t>(std::min<int64_t>(kPageTile, n - b)),
                                 static_cast<int32_t>(b)});
      }
    }
    firstTile[numPages] = static_cast<int64_t>(tiles.size());
  }
  void upload() {
    PageTile* d = static_cast<PageTile*>(tilesBuf.ensure(std::max<size_t>(tiles.size(), 1) * sizeof(PageTile) + 64));
    copyIn(d, tiles.data(), VX355_MEM_HOST, tiles.size() * sizeof(PageTile));
    devTiles = d;
  }
};

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
  // The stream tree, flattened: every scalar column (top level or field of a struct) is one column
  // of 'leaves', loaded like any batch; ROW nodes only bring their null bitmap.
  std::vector<StreamNode> nodes;
  std::vector<int> topLevel;
  std::vector<vx355_column> leaves;
  std::vector<int> leafOfNode;
  std::vector<DevBuf> structNulls;
  structNulls.reserve(static_cast<size_t>(nc));
  auto addLeaf = [&](const vx355_column& col) {
    if (!encodingName(col.type_kind)) {
      VX_THROW(VX355_EUNSUPPORTED, "PrestoPage column of type kind " + std::to_string(col.type_kind));
    }
    StreamNode n;
    n.kind = col.type_kind;
    nodes.push_back(n);
    leafOfNode.push_back(static_cast<int>(leaves.size()));
    leaves.push_back(col);
    return static_cast<int>(nodes.size()) - 1;
  };
  for (int32_t c = 0; c < nc; ++c) {
    const vx355_column& col = batch->cols[c];
    if (col.type_kind != VX355_ROW) {
      topLevel.push_back(addLeaf(col));
      continue;
    }
    VX_CHECK_ARG(col.encoding == VX355_FLAT, "a ROW column is FLAT");
    VX_CHECK_ARG(col.base_size >= 0 && (col.base_size == 0 || col.values), "ROW column without children");
    StreamNode n;
    n.kind = VX355_ROW;
    n.view.kind = kRowNodeKind;
    n.view.enc = VX355_FLAT;
    if (col.nulls) {
      const size_t bytes = static_cast<size_t>(ceilDiv(std::max<int64_t>(batch->num_rows, 1), 64)) * 8;
      if (col.mem == VX355_MEM_HOST) {
        structNulls.emplace_back();
        uint64_t* d = static_cast<uint64_t*>(structNulls.back().ensure(bytes + 64));
        copyIn(d, col.nulls, VX355_MEM_HOST, bytes);
        n.view.nulls = d;
      } else {
        n.view.nulls = col.nulls;
      }
    }
    nodes.push_back(n);
    leafOfNode.push_back(-1);
    const int self = static_cast<int>(nodes.size()) - 1;
    topLevel.push_back(self);
    const auto* kids = static_cast<const vx355_column*>(col.values);
    for (int32_t k = 0; k < col.base_size; ++k) {
      if (kids[k].type_kind == VX355_ROW) {
        VX_THROW(VX355_EUNSUPPORTED, "PrestoPage: a ROW inside a ROW");
      }
      const int child = addLeaf(kids[k]);
      nodes[self].children.push_back(child);
    }
  }
  const int64_t limit = rows ? INT64_MAX : batch->num_rows;
  for (int32_t p = 0; p < numPages; ++p) {
    VX_CHECK_ARG(offsets[p] >= 0 && offsets[p + 1] >= offsets[p] && offsets[p + 1] <= limit, "page row ranges");
    VX_CHECK_ARG(offsets[p + 1] - offsets[p] <= INT32_MAX, "more than 2^31 rows in a page");
  }
  DeviceBatch db;
  {
    std::vector<int32_t> all(leaves.size());
    for (size_t i = 0; i < leaves.size(); ++i) {
      all[i] = static_cast<int32_t>(i);
    }
    vx355_batch leafBatch{batch->num_rows, static_cast<int32_t>(leaves.size()), leaves.data()};
    db.load(&leafBatch, all);
  }
  for (size_t i = 0; i < nodes.size(); ++i) {
    if (leafOfNode[i] >= 0) {
      nodes[i].view = db.col(leafOfNode[i]);
    }
  }
  // group 0: the page rows
  std::vector<std::unique_ptr<RowGroup>> groups;
  groups.push_back(std::make_unique<RowGroup>());
  RowGroup& g0 = *groups[0];
  g0.pageBegin.assign(offsets, offsets + numPages + 1);
  g0.cut(numPages);
  if (g0.tiles.empty()) {
    for (int32_t p = 0; p <= numPages; ++p) {
      pageOffsets[p] = 0;
    }
    return;
  }
  DevBuf dCols, dColTiles, dLaunch, dCounts, dLayout, dPatches, dOut, dFlag, dTileBase;
  if (rows) {
    const int64_t last = offsets[numPages];
    if (rowsMem == VX355_MEM_HOST) {
      int32_t* staged = static_cast<int32_t*>(g0.rowsBuf.ensure(static_cast<size_t>(std::max<int64_t>(last, 1)) * 4 + 64));
      copyIn(staged, rows, VX355_MEM_HOST, static_cast<size_t>(last) * 4);
      g0.devRows = staged;
    } else {
      g0.devRows = rows;
    }
  }
  g0.upload();
  uint32_t* devFlag = static_cast<uint32_t*>(dFlag.ensure(64));
  HIP_OK(hipMemsetAsync(devFlag, 0, 4, rt.stream));
  // cells: one per (stream, tile of its group); a child group has at most as many rows as group 0
  const int64_t maxTiles = static_cast<int64_t>(g0.tiles.size()) + numPages;
  for (size_t i = 0; i < nodes.size(); ++i) {
    nodes[i].cellBase = static_cast<int64_t>(i) * maxTiles;
  }
  const size_t numCells = nodes.size() * static_cast<size_t>(maxTiles);
  uint64_t* devCounts = static_cast<uint64_t*>(dCounts.ensure(std::max<size_t>(numCells, 1) * 16 + 64));
  std::vector<uint64_t> counts(std::max<size_t>(numCells, 1) * 2, 0);
  ColView* devCols = static_cast<ColView*>(dCols.ensure(std::max<size_t>(nodes.size(), 1) * sizeof(ColView) + 64));
  ColTiles* devColTiles = static_cast<ColTiles*>(dColTiles.ensure(std::max<size_t>(nodes.size(), 1) * sizeof(ColTiles) + 64));
  int32_t* devLaunch = static_cast<int32_t*>(dLaunch.ensure(std::max<size_t>(nodes.size(), 1) * 4 + 64));
  PageArgs a{};
  a.cols = devCols;
  a.colTiles = devColTiles;
  a.launchCols = devLaunch;
  a.lossless = lossless ? 1 : 0;
  a.counts = devCounts;
  a.errorFlag = devFlag;
  a.batchRows = batch->num_rows;
  std::vector<ColView> hostViews(nodes.size());
  std::vector<ColTiles> hostColTiles(nodes.size());
  auto uploadStreams = [&]() {
    for (size_t i = 0; i < nodes.size(); ++i) {
      const RowGroup& g = *groups[nodes[i].group];
      hostViews[i] = nodes[i].view;
      hostColTiles[i] = ColTiles{g.devRows, g.devTiles, static_cast<int64_t>(g.tiles.size()), nodes[i].cellBase};
    }
    copyIn(devCols, hostViews.data(), VX355_MEM_HOST, nodes.size() * sizeof(ColView));
    copyIn(devColTiles, hostColTiles.data(), VX355_MEM_HOST, nodes.size() * sizeof(ColTiles));
  };
  // counts of the streams in 'which' (all of one level: their groups exist)
  auto countStreams = [&](const std::vector<int32_t>& which) {
    if (which.empty()) {
      return;
    }
    int64_t gridTiles = 0;
    for (int32_t n : which) {
      gridTiles = std::max<int64_t>(gridTiles, static_cast<int64_t>(groups[nodes[n].group]->tiles.size()));
    }
    if (gridTiles == 0) {
      return;
    }
    rt.sync();  // (the staged launch list of the previous level must have been consumed)
    copyIn(devLaunch, which.data(), VX355_MEM_HOST, which.size() * 4);
    VX_LAUNCH("k_page_count", k_page_count, dim3(static_cast<unsigned>(gridTiles), static_cast<unsigned>(which.size())), 256,
              0, a);
    for (int32_t n : which) {
      const size_t tilesOf = groups[nodes[n].group]->tiles.size();
      if (tilesOf) {
        copyOutAsync(&counts[static_cast<size_t>(nodes[n].cellBase) * 2], VX355_MEM_HOST, devCounts + nodes[n].cellBase * 2,
                     tilesOf * 16);
      }
    }
    rt.sync();
    uint32_t flag = 0;
    copyOut(&flag, VX355_MEM_HOST, devFlag, 4);
    if (flag == 2) {
      VX_THROW(VX355_EINVAL, "rows[] holds a row number outside the batch");
    }
  };
  uploadStreams();
  {
    std::vector<int32_t> level0(topLevel.begin(), topLevel.end());
    countStreams(level0);
  }
  // ROW nodes with a null bitmap: the row list of their children = the rows of the non-null structs
  std::vector<int32_t> level1;
  for (int t : topLevel) {
    StreamNode& node = nodes[t];
    if (node.kind != VX355_ROW) {
      continue;
    }
    if (node.view.nulls != nullptr && !node.children.empty()) {
      groups.push_back(std::make_unique<RowGroup>());
      RowGroup& g = *groups.back();
      const int gid = static_cast<int>(groups.size()) - 1;
      std::vector<uint64_t> tileBase(g0.tiles.size() + 1, 0);
      for (size_t tt = 0; tt < g0.tiles.size(); ++tt) {
        tileBase[tt + 1] = tileBase[tt] + counts[(static_cast<size_t>(node.cellBase) + tt) * 2];
      }
      g.pageBegin.resize(numPages + 1);
      for (int32_t p = 0; p <= numPages; ++p) {
        g.pageBegin[p] = static_cast<int64_t>(tileBase[g0.firstTile[p]]);
      }
      g.cut(numPages);
      int32_t* childRows = static_cast<int32_t*>(g.rowsBuf.ensure(static_cast<size_t>(tileBase.back() + 1) * 4 + 64));
      uint64_t* devTileBase = static_cast<uint64_t*>(dTileBase.ensure(tileBase.size() * 8 + 64));
      rt.sync();
      copyIn(devTileBase, tileBase.data(), VX355_MEM_HOST, tileBase.size() * 8);
      VX_LAUNCH("k_row_compact", k_row_compact, static_cast<int>(g0.tiles.size()), 256, 0, node.view,
                ColTiles{g0.devRows, g0.devTiles, static_cast<int64_t>(g0.tiles.size()), 0}, devTileBase, childRows);
      rt.sync();  // (tileBase is reused by the next ROW column)
      g.devRows = childRows;
      g.upload();
      for (int child : node.children) {
        nodes[child].group = gid;
      }
    }
    for (int child : node.children) {
      level1.push_back(child);
    }
  }
  if (!level1.empty()) {
    uploadStreams();
    countStreams(level1);
  }
  // layout
  std::vector<TileOut> layout(std::max<size_t>(numCells, 1));
  std::vector<PagePatch> patches;
  auto patch = [&](uint64_t pos, const void* data, uint32_t len) {
    PagePatch pp{};
    pp.pos = pos;
    pp.len = len;
    std::memcpy(pp.data, data, len);
    patches.push_back(pp);
  };
  unsigned char word[32];
  // VectorStream::flush of a scalar stream at *pos (default / VARCHAR branches); advances *pos
  auto layoutScalar = [&](const StreamNode& node, int32_t p, int64_t* posInOut) {
    int64_t pos = *posInOut;
    const RowGroup& g = *groups[node.group];
    const int64_t n = g.pageBegin[p + 1] - g.pageBegin[p];
    const char* name = encodingName(node.kind);
    const int32_t nameLen = static_cast<int32_t>(std::strlen(name));
    const bool str = isString(node.kind);
    uint64_t nonNull = 0, bytes = 0;
    for (int64_t t = g.firstTile[p]; t < g.firstTile[p + 1]; ++t) {
      const uint64_t* cell = &counts[static_cast<size_t>(node.cellBase + t) * 2];
      nonNull += cell[0];
      bytes += cell[1];
    }
    if (bytes > INT32_MAX) {
      VX_THROW(VX355_EUSER, "more than 2 GB of string bytes in one page column");
    }
    const bool hasNulls = nonNull < static_cast<uint64_t>(n);
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
    const int w = str ? 1 : valueWidth(node.kind, lossless);
    uint64_t valueRun = 0, byteRun = 0;
    for (int64_t t = g.firstTile[p]; t < g.firstTile[p + 1]; ++t) {
      const size_t cellIndex = static_cast<size_t>(node.cellBase + t);
      TileOut& lo = layout[cellIndex];
      lo.nullPos = nullPos;
      lo.offsetsPos = offsetsPos;
      lo.bytesBefore = byteRun;
      lo.valuePos = static_cast<uint64_t>(pos) + (str ? byteRun : valueRun * w);
      valueRun += counts[cellIndex * 2];
      byteRun += counts[cellIndex * 2 + 1];
    }
    pos += str ? static_cast<int64_t>(bytes) : static_cast<int64_t>(nonNull) * w;
    *posInOut = pos;
  };
  // VectorStream::flush, ROW branch (VectorStream.cpp:236-262): "ROW", the number of fields, the
  // field streams, then the struct's own row count, rows + 1 offsets, null flag and bits
  auto layoutRow = [&](const StreamNode& node, int32_t p, int64_t* posInOut) {
    int64_t pos = *posInOut;
    const int64_t n = g0.pageBegin[p + 1] - g0.pageBegin[p];
    putI32(word, 3);
    std::memcpy(word + 4, "ROW", 3);
    putI32(word + 7, static_cast<int64_t>(node.children.size()));
    patch(static_cast<uint64_t>(pos), word, 11);
    pos += 11;
    for (int child : node.children) {
      layoutScalar(nodes[child], p, &pos);
    }
    uint64_t nonNull = 0;
    for (int64_t t = g0.firstTile[p]; t < g0.firstTile[p + 1]; ++t) {
      nonNull += counts[static_cast<size_t>(node.cellBase + t) * 2];
    }
    const bool hasNulls = nonNull < static_cast<uint64_t>(n);
    putI32(word, n);
    putI32(word + 4, 0);  // offsets[0] (VectorStream::clear, :317-324)
    patch(static_cast<uint64_t>(pos), word, 8);
    pos += 8;
    const uint64_t offsetsPos = static_cast<uint64_t>(pos);  // offsets[1 ...]: the running count behind every row
    pos += 4 * n;
    const unsigned char flag = hasNulls ? 1 : 0;
    patch(static_cast<uint64_t>(pos), &flag, 1);
    pos += 1;
    uint64_t nullPos = ~0ULL;
    if (hasNulls) {
      nullPos = static_cast<uint64_t>(pos);
      pos += (n + 7) / 8;
    }
    uint64_t run = 0;
    for (int64_t t = g0.firstTile[p]; t < g0.firstTile[p + 1]; ++t) {
      const size_t cellIndex = static_cast<size_t>(node.cellBase + t);
      TileOut& lo = layout[cellIndex];
      lo.nullPos = nullPos;
      lo.offsetsPos = offsetsPos;
      lo.bytesBefore = run;
      lo.valuePos = 0;  // a ROW node has no bytes of its own
      run += counts[cellIndex * 2 + 1];
    }
    *posInOut = pos;
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
    putI32(word, nc);
    patch(static_cast<uint64_t>(pos), word, 4);
    pos += 4;
    for (int t : topLevel) {
      if (nodes[t].kind == VX355_ROW) {
        layoutRow(nodes[t], p, &pos);
      } else {
        layoutScalar(nodes[t], p, &pos);
      }
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
  if (!nodes.empty()) {
    TileOut* devLayout = static_cast<TileOut*>(dLayout.ensure(layout.size() * sizeof(TileOut) + 64));
    copyIn(devLayout, layout.data(), VX355_MEM_HOST, layout.size() * sizeof(TileOut));
    a.layout = devLayout;
    std::vector<int32_t> every(nodes.size());
    int64_t gridTiles = 0;
    for (size_t i = 0; i < nodes.size(); ++i) {
      every[i] = static_cast<int32_t>(i);
      gridTiles = std::max<int64_t>(gridTiles, static_cast<int64_t>(groups[nodes[i].group]->tiles.size()));
    }
    rt.sync();
    copyIn(devLaunch, every.data(), VX355_MEM_HOST, every.size() * 4);
    VX_LAUNCH("k_page_write", k_page_write, dim3(static_cast<unsigned>(gridTiles), static_cast<unsigned>(nodes.size())), 256,
              0, a);
    rt.sync();  // ('every' is a local: its upload must have happened before it goes)
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

// ---- the other direction: pages an Exchange received -> flat columns in HBM --------------------
// (PrestoVectorSerde::deserialize, serializers/PrestoSerializer.cpp:120-200; column readers in
// serializers/PrestoSerializerDeserializationUtils.cpp: read<T>, readNulls, the VARIABLE_WIDTH
// reader.) The host walks the framing of every page (a few dozen bytes per column plus a popcount
// over the null bytes, which it needs to find where the values end), the page bytes go to HBM
// once, and one launch expands all pages and columns: a lane per output row finds its page, reads
// its null bit, ranks itself among the non-null rows of its page (host prefix per 64 rows + a
// popcount of the bits before it) and moves its value from the packed stream to its row.

struct ReadSection {
  int64_t nullPos;    // device byte offset of the null bytes; -1: the column has no nulls in this page
  int64_t valuePos;   // first value / first string byte
  int64_t offsetsPos; // VARIABLE_WIDTH: the i32 end offsets
  int64_t prefixBase; // index of this page column's first 64-row block in 'prefix'
  // RLE: every row is row 0 of the nested one-row column. DICTIONARY: row r is row indices[r] of
  // the nested dictionary column (VectorStream::flush's CONSTANT / DICTIONARY branches,
  // serializers/VectorStream.cpp:210-232; readers: readConstantVector / readDictionaryVector).
  int64_t indicesPos; // DICTIONARY: the i32 indices; -1 otherwise
  int32_t constant;   // RLE
  int32_t pad;
  // Field of a ROW column: the stream holds one row per NON-NULL struct (readRowVector,
  // PrestoSerializerDeserializationUtils.cpp:1041-1110). outerNullPos: the struct's null bytes
  // (-1: no struct is null, the stream has a row per page row), outerPrefixBase: their 64-row
  // prefix counts. A page row first becomes its rank among the non-null structs.
  int64_t outerNullPos;
  int64_t outerPrefixBase;
};

struct ReadArgs {
  const unsigned char* bytes;    // all pages, back to back
  const int64_t* pageRowBegin;   // numPages + 1
  int32_t numPages;
  int32_t numCols;
  int32_t lossless;
  int32_t pad;
  int64_t totalRows;
  const ReadSection* sections;   // [page * numCols + col]
  const uint32_t* prefix;        // non-null rows before each 64-row block of a page column
  const int32_t* kinds;          // per column
  void* const* values;           // per column, device
  uint64_t* const* nulls;        // per column, device, may hold nullptr
};

__device__ inline void storeBitWord(uint64_t* words, int64_t pos, bool bit) {
  const uint64_t m = ballot(bit);
  if (words && lane() == 0) {
    words[pos >> 6] = m;
  }
}

__global__ __launch_bounds__(256) void k_page_read(ReadArgs a) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (r - lane() >= a.totalRows) {
    return;  // whole wave past the end
  }
  const int col = blockIdx.y;
  const int32_t kind = a.kinds[col];
  const bool active = r < a.totalRows;
  // page of this row
  int32_t lo = 0, hi = a.numPages - 1;
  while (active && lo < hi) {
    const int32_t mid = (lo + hi + 1) >> 1;
    if (a.pageRowBegin[mid] <= r) {
      lo = mid;
    } else {
      hi = mid - 1;
    }
  }
  int64_t lr = active ? r - a.pageRowBegin[lo] : 0;
  const ReadSection sec =
      active ? a.sections[static_cast<int64_t>(lo) * a.numCols + col] : ReadSection{-1, 0, 0, 0, -1, 0, 0, -1, 0};
  bool structValid = true;
  if (active && sec.outerNullPos >= 0) {
    // field of a struct: my row in the field's stream = my rank among the non-null structs
    const unsigned char* nb = a.bytes + sec.outerNullPos;
    structValid = !((nb[lr >> 3] >> (7 - (lr & 7))) & 1);
    uint64_t rank0 = a.prefix[sec.outerPrefixBase + (lr >> 6)];
    const int64_t blockFirst = lr & ~63LL;
    for (int64_t b = blockFirst >> 3; b < (lr >> 3); ++b) {
      rank0 += 8 - __popc(static_cast<uint32_t>(nb[b]));
    }
    const uint32_t before = static_cast<uint32_t>(nb[lr >> 3]) >> (8 - (lr & 7));
    rank0 += (lr & 7) - __popc((lr & 7) ? before : 0u);
    lr = structValid ? static_cast<int64_t>(rank0) : 0;
  }
  if (sec.constant || !structValid) {
    lr = 0;
  } else if (sec.indicesPos >= 0) {
    lr = static_cast<int64_t>(reinterpret_cast<const Packed32*>(a.bytes + sec.indicesPos + 4 * lr)->v);
  }
  bool valid = active && structValid;
  uint64_t rank = static_cast<uint64_t>(lr);
  if (valid && sec.nullPos >= 0) {
    const unsigned char* nb = a.bytes + sec.nullPos;
    valid = !((nb[lr >> 3] >> (7 - (lr & 7))) & 1);  // first row in the most significant bit, 1 = null
    // rank among the non-null rows: rows of earlier 64-row blocks (host), then the bits before mine
    rank = a.prefix[sec.prefixBase + (lr >> 6)];
    const int64_t blockFirst = lr & ~63LL;
    for (int64_t b = blockFirst >> 3; b < (lr >> 3); ++b) {
      rank += 8 - __popc(static_cast<uint32_t>(nb[b]));
    }
    const uint32_t before = static_cast<uint32_t>(nb[lr >> 3]) >> (8 - (lr & 7));  // the (lr & 7) bits above mine
    rank += (lr & 7) - __popc((lr & 7) ? before : 0u);
  }
  storeBitWord(a.nulls[col], r, valid);
  if (kind == VX355_ROW) {
    return;  // a struct column's own stream: its validity, nothing else
  }
  if (kind == VX355_BOOLEAN) {
    const bool v = valid && a.bytes[sec.valuePos + rank] != 0;
    storeBitWord(static_cast<uint64_t*>(a.values[col]), r, v);
    return;
  }
  if (!active) {
    return;
  }
  const unsigned char* src = a.bytes + sec.valuePos;
  switch (kind) {
    case VX355_TINYINT:
      static_cast<unsigned char*>(a.values[col])[r] = valid ? src[rank] : 0;
      break;
    case VX355_SMALLINT:
      static_cast<uint16_t*>(a.values[col])[r] = valid ? reinterpret_cast<const Packed16*>(src + rank * 2)->v : 0;
      break;
    case VX355_INTEGER:
    case VX355_REAL:
      static_cast<uint32_t*>(a.values[col])[r] = valid ? reinterpret_cast<const Packed32*>(src + rank * 4)->v : 0;
      break;
    case VX355_BIGINT:
    case VX355_DOUBLE:
      static_cast<uint64_t*>(a.values[col])[r] = valid ? reinterpret_cast<const Packed64*>(src + rank * 8)->v : 0;
      break;
    case VX355_TIMESTAMP: {
      int64_t seconds = 0;
      uint64_t nanos = 0;
      if (valid && a.lossless) {
        seconds = static_cast<int64_t>(reinterpret_cast<const Packed64*>(src + rank * 16)->v);
        nanos = reinterpret_cast<const Packed64*>(src + rank * 16 + 8)->v;
      } else if (valid) {
        // Timestamp::fromMillis (type/Timestamp.h:228-235)
        const int64_t ms = static_cast<int64_t>(reinterpret_cast<const Packed64*>(src + rank * 8)->v);
        if (ms >= 0 || ms % 1000 == 0) {
          seconds = ms / 1000;
          nanos = static_cast<uint64_t>(ms % 1000) * 1000000;
        } else {
          seconds = ms / 1000 - 1;
          nanos = static_cast<uint64_t>((ms - seconds * 1000) % 1000) * 1000000;
        }
      }
      static_cast<int64_t*>(a.values[col])[r * 2] = seconds;
      static_cast<uint64_t*>(a.values[col])[r * 2 + 1] = nanos;
      break;
    }
    default: {  // VARCHAR / VARBINARY: end offsets per row, nulls repeat the previous one
      const unsigned char* ends = a.bytes + sec.offsetsPos;
      const uint32_t end = reinterpret_cast<const Packed32*>(ends + 4 * lr)->v;
      const uint32_t begin = lr ? reinterpret_cast<const Packed32*>(ends + 4 * (lr - 1))->v : 0;
      const uint32_t size = valid ? end - begin : 0;
      const unsigned char* data = src + begin;
      uint4 view = make_uint4(size, 0, 0, 0);
      if (size <= 12) {
        uint32_t w[3] = {0, 0, 0};
        for (uint32_t i = 0; i < size; ++i) {
          w[i >> 2] |= static_cast<uint32_t>(data[i]) << ((i & 3) * 8);
        }
        view.y = w[0];
        view.z = w[1];
        view.w = w[2];
      } else {
        view.y = static_cast<uint32_t>(data[0]) | (static_cast<uint32_t>(data[1]) << 8) |
            (static_cast<uint32_t>(data[2]) << 16) | (static_cast<uint32_t>(data[3]) << 24);
        const uint64_t p = reinterpret_cast<uint64_t>(data);
        view.z = static_cast<uint32_t>(p);
        view.w = static_cast<uint32_t>(p >> 32);
      }
      static_cast<uint4*>(a.values[col])[r] = view;
    }
  }
}

int32_t getI32(const unsigned char* p) {
  int32_t v;
  std::memcpy(&v, p, 4);
  return v;
}

// computeChecksum (PrestoSerializer.cpp:39-79): the stored body, then codec, numRows, uncompressedSize.
int64_t pageChecksum(const Crc32& crc, const unsigned char* body, size_t stored, unsigned char codec, int32_t numRows,
                     int32_t uncompressed) {
  uint32_t state = ~0u;
  state = crc.update(state, body, stored);
  state = crc.update(state, &codec, 1);
  state = crc.update(state, reinterpret_cast<const unsigned char*>(&numRows), 4);
  state = crc.update(state, reinterpret_cast<const unsigned char*>(&uncompressed), 4);
  return static_cast<int64_t>(static_cast<uint32_t>(~state));
}

void putI32At(unsigned char* p, int32_t v) { std::memcpy(p, &v, 4); }

// What the reader does with a compressed page before parsing it (PrestoSerializer.cpp:185-199):
// checksum over the bytes as stored, then codec->uncompress(body, uncompressedSize). 'out' becomes the
// page a non-compressing writer would have sent (marker without the compressed bit, size =
// uncompressedSize, checksum - if the page carries one - over the new body).
void uncompressPage(const unsigned char* page, int64_t size, int32_t kind, const std::string& who,
                    std::vector<unsigned char>& out) {
  static const Crc32 crc;
  auto bad = [&](const std::string& what) { VX_THROW(VX355_EUSER, who + ": " + what); };
  if (!page || size < kPageHeader) {
    bad(std::to_string(size) + " bytes for header");
  }
  const int32_t n = getI32(page);
  const unsigned char codec = page[4];
  const int32_t uncompressed = getI32(page + 5);
  const int32_t stored = getI32(page + 9);
  if (n < 0 || uncompressed < 0 || stored < 0) {
    bad("negative header field");
  }
  if (codec & 2) {
    VX_THROW(VX355_EUNSUPPORTED, "encrypted PrestoPage");
  }
  if (static_cast<int64_t>(stored) + kPageHeader != size) {
    bad("size fields do not match the page length");
  }
  if (!(codec & 1)) {
    out.assign(page, page + size);
    return;
  }
  if (kind == VX355_COMPRESSION_NONE) {
    VX_THROW(VX355_EINVAL, who + " is compressed but the flags name no compression kind (VX355_PAGE_COMPRESSION)");
  }
  if (stored >= uncompressed) {
    bad("compressed size is not below the uncompressed size");   // VELOX_CHECK_LT, PrestoSerializer.cpp:48
  }
  if (codec & 4) {
    int64_t expected;
    std::memcpy(&expected, page + 13, 8);
    if (expected != pageChecksum(crc, page + kPageHeader, static_cast<size_t>(stored), codec, n, uncompressed)) {
      bad("Received corrupted serialized page.");
    }
  }
  out.resize(static_cast<size_t>(kPageHeader) + static_cast<size_t>(uncompressed));
  codecUncompress(kind, page + kPageHeader, static_cast<size_t>(stored), out.data() + kPageHeader,
                  static_cast<size_t>(uncompressed));
  const unsigned char plain = static_cast<unsigned char>(codec & ~1u);
  putI32At(out.data(), n);
  out[4] = plain;
  putI32At(out.data() + 5, uncompressed);
  putI32At(out.data() + 9, uncompressed);
  int64_t sum = 0;
  if (plain & 4) {
    sum = pageChecksum(crc, out.data() + kPageHeader, static_cast<size_t>(uncompressed), plain, n, uncompressed);
  }
  std::memcpy(out.data() + 13, &sum, 8);
}

// flushCompressed (PrestoSerializerSerializationUtils.h:279-334) for one finished page.
void compressPage(const unsigned char* page, int64_t size, int32_t kind, float minRatio, std::vector<unsigned char>& out) {
  static const Crc32 crc;
  VX_CHECK_ARG(page && size >= kPageHeader, "a page is at least its 21-byte header");
  const int32_t n = getI32(page);
  const unsigned char codec = page[4];
  const int32_t uncompressed = getI32(page + 5);
  const int32_t stored = getI32(page + 9);
  VX_CHECK_ARG(!(codec & 3), "the page is already compressed or encrypted");
  VX_CHECK_ARG(n >= 0 && uncompressed == stored && static_cast<int64_t>(stored) + kPageHeader == size,
               "size fields do not match the page length");
  if (kind == VX355_COMPRESSION_NONE) {
    out.assign(page, page + size);
    return;
  }
  if (!codecName(kind)) {
    VX_THROW(VX355_EUNSUPPORTED, "compression kind " + std::to_string(kind) + " has no folly codec (Compression.cpp:43)");
  }
  std::vector<unsigned char> body;
  codecCompress(kind, page + kPageHeader, static_cast<size_t>(stored), body);
  if (static_cast<double>(body.size()) > static_cast<double>(uncompressed) * static_cast<double>(minRatio) ||
      body.size() >= static_cast<size_t>(uncompressed)) {
    out.assign(page, page + size);   // not worth it: the page travels uncompressed (:314-323)
    return;
  }
  out.resize(kPageHeader + body.size());
  const unsigned char marked = static_cast<unsigned char>(codec | 1);
  putI32At(out.data(), n);
  out[4] = marked;
  putI32At(out.data() + 5, uncompressed);
  putI32At(out.data() + 9, static_cast<int32_t>(body.size()));
  int64_t sum = 0;
  if (marked & 4) {
    sum = pageChecksum(crc, body.data(), body.size(), marked, n, uncompressed);
  }
  std::memcpy(out.data() + 13, &sum, 8);
  std::memcpy(out.data() + kPageHeader, body.data(), body.size());
}

void deserializePages(const void* const* pagesIn, const int64_t* sizesIn, int32_t numPages, const int32_t* types,
                      int32_t numCols, int32_t flags, void* deviceBytes, int64_t deviceCapacity, vx355_out_column* cols,
                      int64_t capacityRows, int64_t* rowsOut) {
  auto& rt = Runtime::get();
  VX_CHECK_ARG(rowsOut && numPages >= 0 && numCols >= 0, "NULL argument");
  VX_CHECK_ARG(numPages == 0 || (pagesIn && sizesIn), "NULL argument");
  VX_CHECK_ARG(numCols == 0 || (types && cols), "NULL argument");
  const bool lossless = (flags & VX355_PAGE_LOSSLESS_TIMESTAMP) != 0;
  // Compressed pages are uncompressed on the host first (PrestoSerializer.cpp:185-199); from here on
  // 'pages' / 'sizes' name the uncompressed images.
  std::vector<const void*> pages(pagesIn, pagesIn + numPages);
  std::vector<int64_t> sizes(sizesIn, sizesIn + numPages);
  std::vector<std::vector<unsigned char>> uncompressedPages;
  for (int32_t p = 0; p < numPages; ++p) {
    const unsigned char* page = static_cast<const unsigned char*>(pages[p]);
    if (page && sizes[p] >= kPageHeader && (page[4] & 1) && !(page[4] & 2)) {
      uncompressedPages.emplace_back();
      uncompressPage(page, sizes[p], VX355_PAGE_COMPRESSION_OF(flags), "PrestoPage " + std::to_string(p),
                     uncompressedPages.back());
      pages[p] = uncompressedPages.back().data();
      sizes[p] = static_cast<int64_t>(uncompressedPages.back().size());
    }
  }
  static const Crc32 crc;
  std::vector<int64_t> pageRowBegin(numPages + 1, 0), pageDevBegin(numPages + 1, 0);
  std::vector<ReadSection> sections(static_cast<size_t>(numPages) * std::max(numCols, 1));
  std::vector<uint32_t> prefix;
  for (int32_t p = 0; p < numPages; ++p) {
    const unsigned char* page = static_cast<const unsigned char*>(pages[p]);
    const int64_t size = sizes[p];
    auto bad = [&](const std::string& what) {
      VX_THROW(VX355_EUSER, "PrestoPage " + std::to_string(p) + ": " + what);  // PrestoHeader::read / the readers' checks
    };
    if (!page || size < kPageHeader + 4) {
      bad(std::to_string(size) + " bytes for header");
    }
    const int32_t n = getI32(page);
    const unsigned char codec = page[4];
    const int32_t uncompressed = getI32(page + 5);
    const int32_t stored = getI32(page + 9);
    if (n < 0 || uncompressed < 0 || stored < 0) {
      bad("negative header field");
    }
    if (codec & 3) {
      VX_THROW(VX355_EUNSUPPORTED, "encrypted PrestoPage");   // (compressed ones were uncompressed above)
    }
    if (uncompressed != stored || static_cast<int64_t>(stored) + kPageHeader != size) {
      bad("size fields do not match the page length");
    }
    if (codec & 4) {
      // computeChecksum (PrestoSerializer.cpp:39-79)
      uint32_t state = ~0u;
      state = crc.update(state, page + kPageHeader, static_cast<size_t>(stored));
      state = crc.update(state, page + 4, 1);
      state = crc.update(state, page, 4);
      state = crc.update(state, page + 5, 4);
      int64_t expected;
      std::memcpy(&expected, page + 13, 8);
      if (expected != static_cast<int64_t>(static_cast<uint32_t>(~state))) {
        bad("Received corrupted serialized page.");
      }
    }
    pageRowBegin[p + 1] = pageRowBegin[p] + n;
    pageDevBegin[p + 1] = pageDevBegin[p] + size;
    int64_t pos = kPageHeader + 4;  // (the column count is checked against the top-level entries of types[] below)
    // One column stream at 'pos' into section 'c'. expectRows: the rows it must hold; -1 = a field of
    // a struct (its row count = the non-null structs, checked by the caller). Returns its row count.
    auto parseScalar = [&](int32_t c, int64_t expectRows) -> int64_t {
      const char* name = encodingName(types[c]);
      if (!name) {
        VX_THROW(VX355_EUNSUPPORTED, "PrestoPage column of type kind " + std::to_string(types[c]));
      }
      ReadSection& sec = sections[static_cast<size_t>(p) * numCols + c];
      sec.nullPos = -1;
      sec.offsetsPos = 0;
      sec.indicesPos = -1;
      sec.constant = 0;
      sec.outerNullPos = -1;
      sec.outerPrefixBase = 0;
      auto nameIs = [&](const char* what) {
        const int32_t len = static_cast<int32_t>(std::strlen(what));
        return pos + 4 + len <= size && getI32(page + pos) == len && std::memcmp(page + pos + 4, what, len) == 0;
      };
      int64_t outerRows = expectRows;  // rows of the stream as its consumer sees it
      int64_t flatRows = expectRows;   // rows of the flat column that holds the values
      int64_t dictTail = -1;  // DICTIONARY: where the indices start is known after the nested column
      if (nameIs("RLE")) {
        pos += 4 + 3;
        if (pos + 4 > size || (expectRows >= 0 ? getI32(page + pos) != expectRows : getI32(page + pos) < 0)) {
          bad("column " + std::to_string(c) + " run length");
        }
        outerRows = getI32(page + pos);
        pos += 4;
        sec.constant = 1;
        flatRows = 1;
      } else if (nameIs("DICTIONARY")) {
        pos += 4 + 10;
        if (pos + 4 > size || (expectRows >= 0 ? getI32(page + pos) != expectRows : getI32(page + pos) < 0)) {
          bad("column " + std::to_string(c) + " row count");
        }
        outerRows = getI32(page + pos);
        pos += 4;
        dictTail = 0;
        flatRows = -1;  // read from the nested column's own header
      }
      const int32_t nameLen = static_cast<int32_t>(std::strlen(name));
      if (!nameIs(name)) {
        bad("column " + std::to_string(c) + " is not " + name);  // the reader's encoding check
      }
      pos += 4 + nameLen;
      if (pos + 4 > size) {
        bad("truncated column " + std::to_string(c));
      }
      const int64_t rowsHere = getI32(page + pos);
      if (flatRows >= 0 ? rowsHere != flatRows : rowsHere < 0) {
        bad("column " + std::to_string(c) + " row count");
      }
      flatRows = rowsHere;
      if (outerRows < 0) {
        outerRows = rowsHere;
      }
      pos += 4;
      const bool str = isString(types[c]);
      int64_t lastEnd = 0;
      if (str) {
        if (pos + 4LL * flatRows > size) {
          bad("truncated offsets of column " + std::to_string(c));
        }
        // the kernel trusts these: ascending, and (checked below) ending at the byte count
        for (int64_t i = 0; i < flatRows; ++i) {
          const int64_t end = getI32(page + pos + 4 * i);
          if (end < lastEnd) {
            bad("descending string offsets in column " + std::to_string(c));
          }
          lastEnd = end;
        }
        sec.offsetsPos = pageDevBegin[p] + pos;
        pos += 4LL * flatRows;
      }
      if (pos + 1 > size) {
        bad("truncated column " + std::to_string(c));
      }
      const bool hasNulls = page[pos] != 0;
      pos += 1;
      if (hasNulls && cols[c].nulls == nullptr) {
        VX_THROW(VX355_EINVAL, "page " + std::to_string(p) + " carries nulls in column " + std::to_string(c) +
                                   " but the output column has no null buffer");
      }
      int64_t nonNull = flatRows;
      sec.prefixBase = static_cast<int64_t>(prefix.size());
      if (hasNulls) {
        const int64_t nullBytes = (flatRows + 7) / 8;
        if (pos + nullBytes > size) {
          bad("truncated null flags of column " + std::to_string(c));
        }
        sec.nullPos = pageDevBegin[p] + pos;
        uint32_t run = 0;
        for (int64_t b = 0; b < nullBytes; ++b) {
          if ((b & 7) == 0) {
            prefix.push_back(run);  // non-null rows before this 64-row block
          }
          const int64_t inByte = std::min<int64_t>(8, flatRows - b * 8);
          // rows sit in the byte's top bits; pad bits of the last byte are not rows, whatever they hold
          const unsigned flags = page[pos + b] & (0xffu << (8 - inByte)) & 0xffu;
          run += static_cast<uint32_t>(inByte - __builtin_popcount(flags));
        }
        nonNull = run;
        pos += nullBytes;
      }
      int64_t valueBytes;
      if (str) {
        if (pos + 4 > size) {
          bad("truncated column " + std::to_string(c));
        }
        valueBytes = getI32(page + pos);
        pos += 4;
        if (valueBytes != lastEnd) {
          bad("string offsets of column " + std::to_string(c) + " do not end at its byte count");
        }
      } else {
        valueBytes = nonNull * valueWidth(types[c], lossless);
      }
      sec.valuePos = pageDevBegin[p] + pos;
      pos += valueBytes;
      if (valueBytes < 0 || pos > size) {
        bad("truncated values of column " + std::to_string(c));
      }
      if (dictTail == 0) {
        // indices, then 24 bytes of instance id
        if (pos + 4LL * outerRows + 24 > size) {
          bad("truncated dictionary indices of column " + std::to_string(c));
        }
        for (int64_t i = 0; i < outerRows; ++i) {
          const int32_t idx = getI32(page + pos + 4 * i);
          if (idx < 0 || idx >= flatRows) {
            bad("dictionary index out of range in column " + std::to_string(c));
          }
        }
        sec.indicesPos = pageDevBegin[p] + pos;
        pos += 4LL * outerRows + 24;
      }
      return outerRows;
    };
    int32_t topLevelSeen = 0;
    for (int32_t c = 0; c < numCols; ++topLevelSeen) {
      if ((types[c] & 0xff) != VX355_ROW) {
        parseScalar(c, n);
        ++c;
        continue;
      }
      // VectorStream::flush, ROW branch (VectorStream.cpp:236-262) / readRowVector
      // (PrestoSerializerDeserializationUtils.cpp:1041-1110): "ROW", the number of fields, the fields'
      // streams (one row per non-null struct), the struct's row count, rows + 1 offsets, null flag + bits
      const int32_t fields = types[c] >> 8;
      if (c + 1 + fields > numCols) {
        VX_THROW(VX355_EINVAL, "types[]: a ROW entry announces more fields than follow");
      }
      if (!(pos + 4 + 3 + 4 <= size && getI32(page + pos) == 3 && std::memcmp(page + pos + 4, "ROW", 3) == 0)) {
        bad("column " + std::to_string(c) + " is not ROW");
      }
      pos += 7;
      if (getI32(page + pos) != fields) {
        bad("ROW column " + std::to_string(c) + " has " + std::to_string(getI32(page + pos)) + " fields, expected " +
            std::to_string(fields));
      }
      pos += 4;
      std::vector<int64_t> fieldRows(fields);
      for (int32_t f = 0; f < fields; ++f) {
        if ((types[c + 1 + f] & 0xff) == VX355_ROW) {
          VX_THROW(VX355_EUNSUPPORTED, "PrestoPage: a ROW inside a ROW");
        }
        fieldRows[f] = parseScalar(c + 1 + f, -1);
      }
      if (pos + 4 > size || getI32(page + pos) != n) {
        bad("ROW column " + std::to_string(c) + " row count");
      }
      pos += 4;
      if (pos + 4LL * (n + 1) + 1 > size) {
        bad("truncated offsets of ROW column " + std::to_string(c));
      }
      const int64_t offsetsAt = pos;
      pos += 4LL * (n + 1);
      const bool hasNulls = page[pos] != 0;
      pos += 1;
      ReadSection& sec = sections[static_cast<size_t>(p) * numCols + c];
      sec = ReadSection{-1, 0, 0, static_cast<int64_t>(prefix.size()), -1, 0, 0, -1, 0};
      int64_t nonNull = n;
      if (hasNulls) {
        if (cols[c].nulls == nullptr) {
          VX_THROW(VX355_EINVAL, "page " + std::to_string(p) + " carries null structs in column " + std::to_string(c) +
                                     " but the output column has no null buffer");
        }
        const int64_t nullBytes = (n + 7) / 8;
        if (pos + nullBytes > size) {
          bad("truncated null flags of ROW column " + std::to_string(c));
        }
        sec.nullPos = pageDevBegin[p] + pos;
        uint32_t run = 0;
        for (int64_t b = 0; b < nullBytes; ++b) {
          if ((b & 7) == 0) {
            prefix.push_back(run);
          }
          const int64_t inByte = std::min<int64_t>(8, n - b * 8);
          const unsigned flags = page[pos + b] & (0xffu << (8 - inByte)) & 0xffu;
          run += static_cast<uint32_t>(inByte - __builtin_popcount(flags));
        }
        nonNull = run;
        pos += nullBytes;
      }
      // the offsets count the non-null structs: 0, then + 1 behind every non-null row
      if (getI32(page + offsetsAt) != 0 || getI32(page + offsetsAt + 4 * n) != nonNull) {
        bad("offsets of ROW column " + std::to_string(c) + " do not count its non-null rows");
      }
      for (int32_t f = 0; f < fields; ++f) {
        if (fieldRows[f] != nonNull) {
          bad("field " + std::to_string(f) + " of ROW column " + std::to_string(c) + " does not hold one row per non-null struct");
        }
        ReadSection& child = sections[static_cast<size_t>(p) * numCols + c + 1 + f];
        child.outerNullPos = sec.nullPos;
        child.outerPrefixBase = sec.prefixBase;
        if (hasNulls && cols[c + 1 + f].nulls == nullptr) {
          VX_THROW(VX355_EINVAL, "a field of a ROW column with null structs needs a null buffer");
        }
      }
      c += 1 + fields;
    }
    if (getI32(page + kPageHeader) != topLevelSeen) {
      bad("column count " + std::to_string(getI32(page + kPageHeader)) + ", expected " + std::to_string(topLevelSeen));
    }
    if (pos != size) {
      bad("trailing bytes");
    }
  }
  const int64_t totalRows = pageRowBegin[numPages];
  *rowsOut = totalRows;
  if (totalRows == 0 || numCols == 0) {
    return;
  }
  VX_CHECK_ARG(totalRows <= capacityRows, "output columns smaller than the pages' rows");
  VX_CHECK_ARG(deviceBytes && deviceCapacity >= pageDevBegin[numPages], "device buffer smaller than the pages");
  for (int32_t c = 0; c < numCols; ++c) {
    const bool rowNode = (types[c] & 0xff) == VX355_ROW;   // a struct's own entry only carries its null bitmap
    VX_CHECK_ARG(cols[c].type_kind == (types[c] & 0xff) && cols[c].mem == VX355_MEM_DEVICE && (rowNode || cols[c].values),
                 "output columns: device memory of the expected types");
  }
  for (int32_t p = 0; p < numPages; ++p) {
    copyIn(static_cast<char*>(deviceBytes) + pageDevBegin[p], pages[p], VX355_MEM_HOST, static_cast<size_t>(sizes[p]));
  }
  DevBuf dRows, dSections, dPrefix, dKinds, dValues, dNulls;
  int64_t* devRows = static_cast<int64_t*>(dRows.ensure(pageRowBegin.size() * 8 + 64));
  copyIn(devRows, pageRowBegin.data(), VX355_MEM_HOST, pageRowBegin.size() * 8);
  ReadSection* devSections = static_cast<ReadSection*>(dSections.ensure(sections.size() * sizeof(ReadSection) + 64));
  copyIn(devSections, sections.data(), VX355_MEM_HOST, sections.size() * sizeof(ReadSection));
  uint32_t* devPrefix = static_cast<uint32_t*>(dPrefix.ensure(prefix.size() * 4 + 64));
  if (!prefix.empty()) {
    copyIn(devPrefix, prefix.data(), VX355_MEM_HOST, prefix.size() * 4);
  }
  int32_t* devKinds = static_cast<int32_t*>(dKinds.ensure(static_cast<size_t>(numCols) * 4 + 64));
  std::vector<int32_t> kinds(numCols);
  for (int32_t c = 0; c < numCols; ++c) {
    kinds[c] = types[c] & 0xff;
  }
  copyIn(devKinds, kinds.data(), VX355_MEM_HOST, static_cast<size_t>(numCols) * 4);
  std::vector<void*> values(numCols);
  std::vector<uint64_t*> nulls(numCols);
  for (int32_t c = 0; c < numCols; ++c) {
    values[c] = cols[c].values;
    nulls[c] = cols[c].nulls;
  }
  void** devValues = static_cast<void**>(dValues.ensure(static_cast<size_t>(numCols) * 8 + 64));
  uint64_t** devNulls = static_cast<uint64_t**>(dNulls.ensure(static_cast<size_t>(numCols) * 8 + 64));
  copyIn(devValues, values.data(), VX355_MEM_HOST, static_cast<size_t>(numCols) * 8);
  copyIn(devNulls, nulls.data(), VX355_MEM_HOST, static_cast<size_t>(numCols) * 8);
  ReadArgs a{};
  a.bytes = static_cast<const unsigned char*>(deviceBytes);
  a.pageRowBegin = devRows;
  a.numPages = numPages;
  a.numCols = numCols;
  a.lossless = lossless ? 1 : 0;
  a.totalRows = totalRows;
  a.sections = devSections;
  a.prefix = devPrefix;
  a.kinds = devKinds;
  a.values = devValues;
  a.nulls = devNulls;
  VX_LAUNCH("k_page_read", k_page_read,
            dim3(static_cast<unsigned>(ceilDiv(totalRows, 256)), static_cast<unsigned>(numCols)), 256, 0, a);
  rt.sync();
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

int vx355_presto_compress_page(const void* page, int64_t size, int32_t compression, float min_ratio, void* out,
                               int64_t out_capacity, int64_t* out_size) {
  try {   // host work only: no execution context, no GPU
    VX_CHECK_ARG(page && out && out_size, "NULL argument");
    VX_CHECK_ARG(min_ratio > 0, "min_ratio must be positive (PrestoOptions::minCompressionRatio)");
    std::vector<unsigned char> result;
    vx::compressPage(static_cast<const unsigned char*>(page), size, compression, min_ratio, result);
    VX_CHECK_ARG(out_capacity >= static_cast<int64_t>(result.size()), "output buffer smaller than the page");
    std::memcpy(out, result.data(), result.size());
    *out_size = static_cast<int64_t>(result.size());
  VX_API_CATCH
}

int vx355_presto_uncompress_page(const void* page, int64_t size, int32_t compression, void* out, int64_t out_capacity,
                                 int64_t* out_size) {
  try {
    VX_CHECK_ARG(page && out && out_size, "NULL argument");
    std::vector<unsigned char> result;
    vx::uncompressPage(static_cast<const unsigned char*>(page), size, compression, "PrestoPage", result);
    VX_CHECK_ARG(out_capacity >= static_cast<int64_t>(result.size()), "output buffer smaller than the uncompressed page");
    std::memcpy(out, result.data(), result.size());
    *out_size = static_cast<int64_t>(result.size());
  VX_API_CATCH
}

int vx355_presto_deserialize(const void* const* pages, const int64_t* sizes, int32_t num_pages, const int32_t* types,
                             int32_t num_cols, int32_t flags, void* device_bytes, int64_t device_bytes_capacity,
                             vx355_out_column* cols, int64_t capacity_rows, int64_t* rows_out) {
  VX_API_BEGIN
  vx::Runtime::get().requireInit();
  vx::deserializePages(pages, sizes, num_pages, types, num_cols, flags, device_bytes, device_bytes_capacity, cols,
                       capacity_rows, rows_out);
  VX_API_END
}

}  // extern "C"
