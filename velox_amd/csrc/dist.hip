// Multi-GPU orchestration behind the C ABI: what a host that drives the GPUs
// of one node calls instead of re-implementing the plan fragments around the exchange.
//
//   vx355_join_repartition   = PartitionedOutput(keys) -> Exchange -> HashBuild on the build
//                              side, then PartitionedOutput(keys) -> Exchange -> HashProbe on the
//                              probe side, chunk-pipelined: the slices of chunk i ride the xGMI
//                              links while chunk i + 1 is hashed and grouped and chunk i - 1 is
//                              probed (exec/HashPartitionFunction.cpp:76-118,
//                              exec/PartitionedOutput.cpp:59-133, exec/Exchange.cpp,
//                              exec/HashBuild.cpp:442-598, exec/HashProbe.cpp:796-900).
//   vx355_agg_merge_partials = partial aggregation -> PartitionedOutput (PrestoPages) ->
//                              Exchange (all ranks see all pages) -> final aggregation
//                              (docs/develop/aggregations.rst:24-91).
//
// Both are compositions of the library's own entry points (exchange.hip, join.hip, agg.hip,
// serializer.hip); nothing here touches operator internals.
#include "common.h"

#include <algorithm>
#include <array>

#include <algorithm>

using namespace vx;

extern "C" int vx355_all_gather_v(vx355_comm* c, const void* send, const int64_t* sizes, void* recv);

namespace vx {
bool commViaRccl(const vx355_comm* c);  // exchange.hip: false = one rank whose exchanges are local copies
Runtime* commContext(const vx355_comm* c);  // the communicator's execution context (nullptr for nullptr)
}

namespace vx {
namespace {

void ok(int status) {
  if (status != VX355_OK) {
    VX_THROW(status, vx355_last_error());
  }
}

// Rows [begin, end) of a batch of FLAT columns.
std::vector<vx355_column> sliceColumns(const vx355_batch* batch, int64_t begin) {
  std::vector<vx355_column> cols(batch->cols, batch->cols + batch->num_cols);
  for (auto& col : cols) {
    if (col.encoding != VX355_FLAT || col.nulls) {
      VX_THROW(VX355_EUNSUPPORTED, "exchange columns must be FLAT without nulls");
    }
    const int w = kindWidth(col.type_kind);
    if (w <= 0) {
      VX_THROW(VX355_EUNSUPPORTED, "exchange column of type kind " + std::to_string(col.type_kind));
    }
    col.values = static_cast<const char*>(col.values) + begin * w;
  }
  return cols;
}

struct ExchangeHandle {
  vx355_exchange* x = nullptr;
  ~ExchangeHandle() { vx355_exchange_destroy(x); }
};
struct BuildHandle {
  vx355_join_build* b = nullptr;
  ~BuildHandle() { vx355_join_build_destroy(b); }
};
struct ProbeHandle {
  vx355_join_probe* p = nullptr;
  ~ProbeHandle() { vx355_join_probe_destroy(p); }
};
struct TableRef {
  vx355_join_table* t = nullptr;
  ~TableRef() {
    if (t) {
      vx355_join_table_release(t);
    }
  }
};
struct AggHandle {
  vx355_agg* a = nullptr;
  ~AggHandle() { vx355_agg_destroy(a); }
};

std::vector<int32_t> typesOf(const vx355_batch* b) {
  std::vector<int32_t> t(b->num_cols);
  for (int32_t i = 0; i < b->num_cols; ++i) {
    t[i] = b->cols[i].type_kind;
  }
  return t;
}

constexpr int64_t kMaxBatchRows = 1LL << 30;  // vx355_batch.num_rows is a vector_size_t

}  // namespace
}  // namespace vx

extern "C" {

int vx355_join_repartition(vx355_comm* c, const vx355_join_build_spec* build_spec, const vx355_batch* build_rows,
                           const vx355_join_probe_spec* probe_spec, const vx355_batch* probe_rows, int32_t chunks,
                           vx355_join_chunk_sink sink, void* sink_arg, vx355_join_table** table_out) {
  // In the communicator's own execution context, not the device's default one: a default context
  // serialises its entry points (callMutex), and ranks that share a device inside one process
  // (vx355_comm_create_all over one GPU) would wait for each other's collective forever.
  VX_API_BEGIN_CTX(commContext(c))
  VX_CHECK_ARG(c && build_spec && build_rows && probe_spec && probe_rows && sink && table_out, "NULL argument");
  VX_CHECK_ARG(build_rows->num_cols >= 1 && probe_rows->num_cols >= 1, "batches without columns");
  VX_CHECK_ARG(build_spec->num_keys == probe_spec->num_keys && build_spec->num_keys >= 1, "key lists differ");
  int32_t world = 1;
  ok(vx355_comm_info(c, &world, nullptr, nullptr));
  // One rank and no forced RCCL self-exchange: PartitionedOutput -> Exchange over one destination is
  // the identity (the reference's LocalExchange with one partition passes vectors through); the
  // caller's rows, which stay valid for the whole call, feed the operators directly.
  const bool local = world == 1 && !commViaRccl(c);
  // ---- build side: one exchange, then HashBuild over the rows that landed here
  TableRef table;
  {
    ExchangeHandle ex;
    std::vector<vx355_column> got(build_rows->cols, build_rows->cols + build_rows->num_cols);
    int64_t rows = build_rows->num_rows;
    if (!local) {
      const auto types = typesOf(build_rows);
      ok(vx355_exchange_create(c, types.data(), build_rows->num_cols, build_spec->key_cols, build_spec->num_keys, &ex.x));
      ok(vx355_exchange_send(ex.x, build_rows));
      ok(vx355_exchange_receive(ex.x, got.data(), &rows));
    }
    BuildHandle build;
    ok(vx355_join_build_create(build_spec, &build.b));
    int64_t at = 0;
    do {
      vx355_batch piece{static_cast<int32_t>(std::min(kMaxBatchRows, rows - at)), build_rows->num_cols, got.data()};
      const auto cols = sliceColumns(&piece, at);
      piece.cols = cols.data();
      ok(vx355_join_build_add_input(build.b, &piece));
      at += kMaxBatchRows;
    } while (at < rows);
    ok(vx355_join_build_finish(build.b, nullptr, 0, &table.t));
  }
  // ---- probe side, pipelined
  const int64_t n = probe_rows->num_rows;
  // Every chunk is a collective (vx355_exchange_send all-gathers the slice sizes and posts grouped
  // sends / receives): all ranks must run the SAME number of them, whatever their own row counts
  // are - a rank with fewer rows (or none) sends empty chunks. The count is the caller's 'chunks'
  // (the same on every rank by contract), raised to what the largest shard needs to keep a chunk
  // inside a batch's int32 row count; the ranks agree on that with one all-gather of the counts.
  int64_t numChunks = std::max<int64_t>(1, chunks);
  {
    std::vector<int64_t> mine(static_cast<size_t>(world), ceilDiv(n, kMaxBatchRows)), theirs(static_cast<size_t>(world), 0);
    ok(vx355_exchange_counts(c, mine.data(), theirs.data()));
    for (int64_t need : theirs) {
      numChunks = std::max(numChunks, need);
    }
  }
  ExchangeHandle ex;
  const auto types = typesOf(probe_rows);
  if (!local) {
    ok(vx355_exchange_create(c, types.data(), probe_rows->num_cols, probe_spec->key_cols, probe_spec->num_keys, &ex.x));
  }
  ProbeHandle probe;
  ok(vx355_join_probe_create(table.t, probe_spec, &probe.p));
  auto sendChunk = [&](int64_t i) {
    if (local) {
      return;
    }
    const int64_t begin = n * i / numChunks, end = n * (i + 1) / numChunks;
    const auto cols = sliceColumns(probe_rows, begin);
    vx355_batch piece{static_cast<int32_t>(end - begin), probe_rows->num_cols, cols.data()};
    ok(vx355_exchange_send(ex.x, &piece));
  };
  // The operator's input batch is the chunk regrouped by slice of the join table when the table is
  // beyond the caches (vx355_join_probe_add_input_regrouped): the sink sees that batch.
  std::vector<DevBuf> moved(probe_rows->num_cols);
  auto consumeChunk = [&](int64_t i) {
    std::vector<vx355_column> got(probe_rows->num_cols);
    int64_t rows = 0;
    std::vector<vx355_column> sliced;
    if (local) {
      const int64_t begin = n * i / numChunks, end = n * (i + 1) / numChunks;
      sliced = sliceColumns(probe_rows, begin);
      got = sliced;
      rows = end - begin;
    } else {
      ok(vx355_exchange_receive(ex.x, got.data(), &rows));
    }
    if (rows > INT32_MAX) {
      VX_THROW(VX355_EINVAL, "a probe chunk received more than 2^31 rows: ask for more chunks");
    }
    vx355_batch received{static_cast<int32_t>(rows), probe_rows->num_cols, got.data()};
    std::vector<void*> movedPtrs(probe_rows->num_cols);
    for (int32_t col = 0; col < probe_rows->num_cols; ++col) {
      movedPtrs[col] = moved[col].ensure(static_cast<size_t>(std::max<int64_t>(rows, 1)) * kindWidth(types[col]) + 64);
    }
    int32_t regrouped = 0;
    ok(vx355_join_probe_add_input_regrouped(probe.p, &received, movedPtrs.data(), &regrouped));
    if (regrouped) {
      for (int32_t col = 0; col < probe_rows->num_cols; ++col) {
        got[col].values = movedPtrs[col];
        got[col].mem = VX355_MEM_DEVICE;
      }
    } else if (local) {
      // the sink's contract: 'received' holds device columns (an exchange lands them in HBM; here the
      // caller's own rows were probed in place, and host columns are brought over for the sink)
      for (int32_t col = 0; col < probe_rows->num_cols; ++col) {
        if (got[col].mem == VX355_MEM_HOST && rows > 0) {
          copyIn(movedPtrs[col], got[col].values, VX355_MEM_HOST, static_cast<size_t>(rows) * kindWidth(types[col]));
          got[col].values = movedPtrs[col];
          got[col].mem = VX355_MEM_DEVICE;
        }
      }
      Runtime::get().sync();
    }
    const int rc = sink(sink_arg, static_cast<int32_t>(i), &received, probe.p);
    if (rc != VX355_OK) {
      VX_THROW(rc, "vx355_join_repartition: the sink failed on chunk " + std::to_string(i));
    }
  };
  sendChunk(0);
  for (int64_t i = 1; i < numChunks; ++i) {
    sendChunk(i);          // chunk i goes onto the links ...
    consumeChunk(i - 1);   // ... while chunk i - 1 is probed
  }
  consumeChunk(numChunks - 1);
  *table_out = table.t;
  table.t = nullptr;
  VX_API_END
}

int vx355_agg_merge_partials(vx355_comm* c, vx355_agg* partial, const vx355_agg_spec* final_spec,
                             vx355_agg** final_out) {
  VX_API_BEGIN_CTX(commContext(c))  // (see vx355_join_repartition)
  VX_CHECK_ARG(c && partial && final_spec && final_out, "NULL argument");
  VX_CHECK_ARG(final_spec->step == VX355_STEP_FINAL || final_spec->step == VX355_STEP_INTERMEDIATE,
               "the merging operator takes intermediate input (FINAL or INTERMEDIATE step)");
  auto& rt = Runtime::get();
  int32_t world = 1, rank = 0;
  ok(vx355_comm_info(c, &world, &rank, nullptr));
  // ---- this rank's partial groups, as flat device columns
  int32_t numCols = 0;
  ok(vx355_agg_output_types(partial, nullptr, 0, &numCols));
  std::vector<int32_t> types(numCols);
  ok(vx355_agg_output_types(partial, types.data(), numCols, &numCols));
  vx355_agg_stats stats{};
  ok(vx355_agg_get_stats(partial, &stats));
  const int64_t capacity = std::max<int64_t>(64, (stats.num_groups + 63) / 64 * 64);
  std::vector<DevBuf> values(numCols), nulls(numCols);
  std::vector<int> widths(numCols);
  for (int32_t i = 0; i < numCols; ++i) {
    widths[i] = kindWidth(types[i]);
    if (widths[i] < 0) {
      VX_THROW(VX355_EUNSUPPORTED, "partial output column of type kind " + std::to_string(types[i]));
    }
    values[i].ensure(static_cast<size_t>(capacity) * std::max(widths[i], 1) + 64);   // BOOLEAN: bits, over-allocated
    nulls[i].ensure(static_cast<size_t>(capacity / 8) + 64);
  }
  int64_t groups = 0;
  constexpr int32_t kPage = 1 << 24;  // a multiple of 64: page k starts on a null-word boundary
  for (;;) {
    VX_CHECK_ARG(groups < capacity || stats.num_groups == 0, "the partial operator lists more groups than its statistics");
    std::vector<vx355_out_column> page(numCols);
    for (int32_t i = 0; i < numCols; ++i) {
      page[i].type_kind = types[i];
      page[i].mem = VX355_MEM_DEVICE;
      page[i].values = values[i].as<char>() + (widths[i] == 0 ? groups / 8 : groups * widths[i]);
      page[i].nulls = reinterpret_cast<uint64_t*>(nulls[i].as<char>() + groups / 8);
    }
    int32_t got = 0, finished = 0;
    const int32_t room = static_cast<int32_t>(std::min<int64_t>(kPage, capacity - groups));
    ok(vx355_agg_get_output(partial, page.data(), numCols, std::max(room, 1), &got, &finished));
    groups += got;
    if (finished) {
      break;
    }
  }
  // ---- PartitionedOutput: one PrestoPage (the wire format of Velox's own exchange). An avg's
  // (sum, count) pair travels as the reference's intermediate type ROW(DOUBLE, BIGINT)
  // (AverageAggregateBase.h:66-260): a stock Velox FINAL aggregation can consume the page, and a
  // page a Velox PARTIAL wrote can be consumed here. The struct is null where the sum is (a group
  // without non-null inputs).
  const std::vector<int32_t> avgSums = aggPartialAvgColumns(partial);
  auto isAvgSum = [&](int32_t i) { return std::find(avgSums.begin(), avgSums.end(), i) != avgSums.end(); };
  std::vector<vx355_column> cols;
  std::vector<std::array<vx355_column, 2>> fields(avgSums.size());
  std::vector<int32_t> wireTypes;   // prefix order, for the reader
  std::vector<int32_t> nodeOfColumn(numCols, -1);
  size_t pair = 0;
  for (int32_t i = 0; i < numCols; ++i) {
    if (isAvgSum(i)) {
      VX_CHECK_ARG(i + 1 < numCols && types[i] == VX355_DOUBLE && types[i + 1] == VX355_BIGINT, "avg pair layout");
      fields[pair][0] = vx355_column{VX355_DOUBLE, VX355_FLAT, values[i].ptr(), nullptr, nullptr, 0, VX355_MEM_DEVICE};
      fields[pair][1] = vx355_column{VX355_BIGINT, VX355_FLAT, values[i + 1].ptr(), nulls[i + 1].as<uint64_t>(), nullptr, 0,
                                     VX355_MEM_DEVICE};
      cols.push_back(vx355_column{VX355_ROW, VX355_FLAT, fields[pair].data(), nulls[i].as<uint64_t>(), nullptr, 2,
                                  VX355_MEM_DEVICE});
      wireTypes.push_back(VX355_ROW_OF(2));
      nodeOfColumn[i] = static_cast<int32_t>(wireTypes.size());
      wireTypes.push_back(VX355_DOUBLE);
      nodeOfColumn[i + 1] = static_cast<int32_t>(wireTypes.size());
      wireTypes.push_back(VX355_BIGINT);
      ++pair;
      ++i;
      continue;
    }
    cols.push_back(vx355_column{types[i], VX355_FLAT, values[i].ptr(), nulls[i].as<uint64_t>(), nullptr, 0, VX355_MEM_DEVICE});
    nodeOfColumn[i] = static_cast<int32_t>(wireTypes.size());
    wireTypes.push_back(types[i]);
  }
  VX_CHECK_ARG(groups <= INT32_MAX, "more than 2^31 partial groups on one rank");
  vx355_batch mine{static_cast<int32_t>(groups), static_cast<int32_t>(cols.size()), cols.data()};
  const int64_t offsets[2] = {0, groups};
  int64_t pageOffsets[2] = {0, 0};
  const int32_t flags = VX355_PAGE_LOSSLESS_TIMESTAMP;
  ok(vx355_presto_serialize(&mine, nullptr, VX355_MEM_DEVICE, offsets, 1, flags, nullptr, 0, VX355_MEM_DEVICE,
                            pageOffsets));
  const int64_t myBytes = pageOffsets[1];
  DevBuf pageBuf;
  pageBuf.ensure(static_cast<size_t>(myBytes) + 64);
  if (myBytes > 0) {
    ok(vx355_presto_serialize(&mine, nullptr, VX355_MEM_DEVICE, offsets, 1, flags, pageBuf.ptr(), myBytes,
                              VX355_MEM_DEVICE, pageOffsets));
  }
  // ---- Exchange: every rank receives every rank's page
  std::vector<int64_t> sendSizes(world, myBytes), sizes(world, 0);
  ok(vx355_exchange_counts(c, sendSizes.data(), sizes.data()));
  int64_t total = 0;
  for (int64_t s : sizes) {
    total += s;
  }
  DevBuf gathered;
  gathered.ensure(static_cast<size_t>(total) + 64);
  ok(vx355_all_gather_v(c, pageBuf.ptr(), sizes.data(), gathered.ptr()));
  std::vector<char> host(static_cast<size_t>(total));
  copyOut(host.data(), VX355_MEM_HOST, gathered.ptr(), static_cast<size_t>(total));
  std::vector<const void*> pages;
  std::vector<int64_t> pageSizes;
  int64_t rows = 0, at = 0;
  for (int32_t s = 0; s < world; ++s) {
    if (sizes[s] > 0) {
      VX_CHECK_ARG(sizes[s] >= 4, "truncated page");
      int32_t pageRows;
      std::memcpy(&pageRows, host.data() + at, 4);
      rows += pageRows;
      pages.push_back(host.data() + at);
      pageSizes.push_back(sizes[s]);
    }
    at += sizes[s];
  }
  VX_CHECK_ARG(rows <= INT32_MAX, "more than 2^31 partial groups in total");
  // ---- final aggregation over all partial rows, rank order
  std::vector<DevBuf> inValues(numCols), inNulls(numCols);
  std::vector<vx355_out_column> inCols(numCols);
  const int64_t inCap = std::max<int64_t>(rows, 1);
  for (int32_t i = 0; i < numCols; ++i) {
    inCols[i].type_kind = types[i];
    inCols[i].mem = VX355_MEM_DEVICE;
    inCols[i].values = inValues[i].ensure(static_cast<size_t>(inCap) * std::max(widths[i], 1) + 64);
    inCols[i].nulls = static_cast<uint64_t*>(inNulls[i].ensure(static_cast<size_t>(inCap / 8) + 64));
  }
  int64_t rowsOut = 0;
  // the reader's column tree: every flat column, and per avg pair the struct's own entry (its validity
  // is not needed downstream: a field of a null struct comes back null) followed by its two fields
  std::vector<vx355_out_column> wireCols(wireTypes.size());
  DevBuf structNulls;
  structNulls.ensure(static_cast<size_t>(inCap / 8) + 64);
  for (size_t n = 0; n < wireTypes.size(); ++n) {
    if ((wireTypes[n] & 0xff) == VX355_ROW) {
      wireCols[n] = vx355_out_column{VX355_ROW, VX355_MEM_DEVICE, nullptr, structNulls.as<uint64_t>()};
    }
  }
  for (int32_t i = 0; i < numCols; ++i) {
    wireCols[nodeOfColumn[i]] = inCols[i];
  }
  // gathered doubles as the string buffer of the deserialised views (strings > 12 bytes)
  ok(vx355_presto_deserialize(pages.data(), pageSizes.data(), static_cast<int32_t>(pages.size()), wireTypes.data(),
                              static_cast<int32_t>(wireTypes.size()), flags, gathered.ptr(),
                              static_cast<int64_t>(gathered.capacity()), wireCols.data(), inCap, &rowsOut));
  AggHandle fin;
  ok(vx355_agg_create(final_spec, &fin.a));
  std::vector<vx355_column> finCols(numCols);
  for (int32_t i = 0; i < numCols; ++i) {
    finCols[i] = vx355_column{types[i], VX355_FLAT, inCols[i].values, inCols[i].nulls, nullptr, 0, VX355_MEM_DEVICE};
  }
  vx355_batch all{static_cast<int32_t>(rowsOut), numCols, finCols.data()};
  ok(vx355_agg_add_input(fin.a, &all));
  ok(vx355_agg_no_more_input(fin.a));
  rt.sync();
  *final_out = fin.a;
  fin.a = nullptr;
  VX_API_END
}

}  // extern "C"
