// vx355.hpp — header-only C++ host side over the C ABI (vx355.h).
//
// The classes mirror the part of exec::Operator (exec/Operator.h:232-342) the
// three replaced operators implement — needsInput / addInput / noMoreInput /
// getOutput / isFinished / close — with the same call order, so that an
// exec::Operator shim forwards one to one and host programs
// read like the reference's operator tests. Errors surface the way Velox's do:
// VX355_EUSER -> vx355::UserError (VeloxUserError), everything else ->
// vx355::RuntimeError (VeloxRuntimeError). Nothing here touches HIP: all device
// work is behind libvx355.so.
#ifndef VX355_HPP_
#define VX355_HPP_

#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "vx355.h"

namespace vx355 {

struct Error : std::runtime_error {
  Error(int status, const std::string& what) : std::runtime_error(what), status(status) {}
  int status;
};
struct UserError : Error {  // common/base/VeloxException.h:354 VeloxUserError
  using Error::Error;
};
struct RuntimeError : Error {  // :420 VeloxRuntimeError
  using Error::Error;
};

inline void check(int status) {
  if (status == VX355_OK) {
    return;
  }
  const char* msg = vx355_last_error();
  if (status == VX355_EUSER) {
    throw UserError(status, msg ? msg : "user error");
  }
  throw RuntimeError(status, msg ? msg : "vx355 failure");
}

inline void init(int device = 0) { check(vx355_init(device)); }

// exec::HashAggregation (exec/HashAggregation.h:24-118).
class HashAggregation {
 public:
  HashAggregation(std::vector<int32_t> keyChannels, std::vector<int32_t> keyTypes, std::vector<vx355_agg_fn> aggregates,
                  vx355_agg_step step = VX355_STEP_SINGLE, bool ignoreNullKeys = false, int32_t flags = 0)
      : keyChannels_(std::move(keyChannels)), keyTypes_(std::move(keyTypes)), aggregates_(std::move(aggregates)) {
    vx355_agg_spec spec{};
    spec.num_keys = static_cast<int32_t>(keyChannels_.size());
    spec.key_cols = keyChannels_.data();
    spec.key_types = keyTypes_.data();
    spec.num_aggs = static_cast<int32_t>(aggregates_.size());
    spec.aggs = aggregates_.data();
    spec.step = step;
    spec.ignore_null_keys = ignoreNullKeys ? 1 : 0;
    spec.flags = flags;  // VX355_AGG_UNORDERED_OUTPUT
    check(vx355_agg_create(&spec, &handle_));  // VX355_EUNSUPPORTED: keep the CPU operator
  }
  HashAggregation(const HashAggregation&) = delete;
  HashAggregation& operator=(const HashAggregation&) = delete;
  ~HashAggregation() { close(); }

  // Fuse the upstream FilterProject (vx355_agg_set_fused_input).
  void setFusedInput(const std::vector<vx355_filter_term>& terms, const std::vector<vx355_projection>& projections) {
    check(vx355_agg_set_fused_input(handle_, terms.data(), static_cast<int32_t>(terms.size()), projections.data(),
                                    static_cast<int32_t>(projections.size())));
  }

  bool needsInput() const { return !noMoreInput_; }  // HashAggregation.h:50
  void addInput(const vx355_batch& input) { check(vx355_agg_add_input(handle_, &input)); }
  // Asynchronous boundary: the batch is queued for the handle's worker thread; its buffers must stay
  // valid until inFlight() no longer counts its ticket. isBlocked() is what exec::Operator::isBlocked
  // polls; wait() rethrows the first failure among the queued batches.
  int64_t addInputAsync(const vx355_batch& input) {
    int64_t ticket = 0;
    check(vx355_agg_add_input_async(handle_, &input, &ticket));
    return ticket;
  }
  int64_t completedTickets() const {
    int64_t submitted = 0, completed = 0;
    check(vx355_agg_poll(handle_, &submitted, &completed));
    return completed;
  }
  int64_t inFlight() const {
    int64_t submitted = 0, completed = 0;
    check(vx355_agg_poll(handle_, &submitted, &completed));
    return submitted - completed;
  }
  bool isBlocked(int64_t maxInFlight = 2) const { return inFlight() >= maxInFlight; }
  void wait() { check(vx355_agg_wait(handle_)); }
  void noMoreInput() {
    noMoreInput_ = true;
    check(vx355_agg_no_more_input(handle_));
  }
  // Result types in output order (keys, then aggregates; partial avg = sum, count).
  std::vector<int32_t> outputTypes() const {
    std::vector<int32_t> types(64);
    int32_t n = 0;
    check(vx355_agg_output_types(handle_, types.data(), static_cast<int32_t>(types.size()), &n));
    types.resize(n);
    return types;
  }
  // Writes at most maxRows rows into caller-owned columns; returns the row count (0 = nothing left,
  // exec::Operator::getOutput's nullptr).
  int32_t getOutput(vx355_out_column* columns, int32_t numColumns, int32_t maxRows) {
    if (!noMoreInput_ || finished_) {
      return 0;
    }
    int32_t n = 0, finished = 0;
    check(vx355_agg_get_output(handle_, columns, numColumns, maxRows, &n, &finished));
    finished_ = finished != 0;
    return n;
  }
  // Queued forms (ABI 7): noMoreInput and one page of output as tasks behind the queued batches. 'done' (may be
  // null) runs on the library's worker thread when the page is there - fulfil the promise behind the future
  // exec::Operator::isBlocked returned. outputResult(ticket) hands the row count to the Driver thread, once, after
  // completedTickets() >= ticket.
  int64_t noMoreInputAsync() {
    noMoreInput_ = true;
    int64_t ticket = 0;
    check(vx355_agg_no_more_input_async(handle_, &ticket));
    return ticket;
  }
  int64_t getOutputAsync(const vx355_out_column* columns, int32_t numColumns, int32_t maxRows,
                         vx355_output_done_fn done = nullptr, void* doneArg = nullptr) {
    int64_t ticket = 0;
    check(vx355_agg_get_output_async(handle_, columns, numColumns, maxRows, done, doneArg, &ticket));
    return ticket;
  }
  int32_t outputResult(int64_t ticket) {
    int32_t n = 0, finished = 0;
    check(vx355_agg_output_result(handle_, ticket, &n, &finished));
    finished_ = finished != 0;
    return n;
  }
  bool isFinished() const { return finished_; }
  vx355_agg* handle() const { return handle_; }  // for the entry points that take the handle (mergePartials)
  vx355_agg_stats stats() const {
    vx355_agg_stats s{};
    check(vx355_agg_get_stats(handle_, &s));
    return s;
  }
  void close() {
    if (handle_) {
      vx355_agg_destroy(handle_);
      handle_ = nullptr;
    }
  }

 private:
  std::vector<int32_t> keyChannels_, keyTypes_;
  std::vector<vx355_agg_fn> aggregates_;
  vx355_agg* handle_ = nullptr;
  bool noMoreInput_ = false;
  bool finished_ = false;
};

// PartitionedOutput's byte work (exec/PartitionedOutput.cpp:59-133): the rows of every destination
// as PrestoPages. Returns the pages back to back; pageOffsets gets numPages + 1 entries.
// wrapChild over an already wrapped vector (exec/OperatorUtils.cpp:393-422): out[i] = inner[outer[i]].
inline void composeIndices(const int32_t* inner, int32_t innerSize, const int32_t* outer, int32_t numRows, int32_t* out,
                           int32_t mem) {
  check(vx355_compose_indices(inner, innerSize, outer, numRows, out, mem));
}

// GB/s of the library's read-only-stream (VX355_CEILING_READ) or copy (VX355_CEILING_COPY) kernel on this GPU.
inline double hbmCeiling(int32_t kind, size_t bytes, int32_t iterations = 5) {
  double gbps = 0;
  check(vx355_hbm_ceiling(kind, bytes, iterations, &gbps));
  return gbps;
}

inline std::vector<char> prestoSerialize(const vx355_batch& input, const int32_t* rows, int32_t rowsMem,
                                         const std::vector<int64_t>& offsets, int32_t flags,
                                         std::vector<int64_t>* pageOffsets) {
  const int32_t numPages = static_cast<int32_t>(offsets.size()) - 1;
  pageOffsets->assign(offsets.size(), 0);
  check(vx355_presto_serialize(&input, rows, rowsMem, offsets.data(), numPages, flags, nullptr, 0, VX355_MEM_HOST,
                               pageOffsets->data()));
  std::vector<char> out(static_cast<size_t>(pageOffsets->back()));
  check(vx355_presto_serialize(&input, rows, rowsMem, offsets.data(), numPages, flags, out.data(),
                               static_cast<int64_t>(out.size()), VX355_MEM_HOST, pageOffsets->data()));
  return out;
}

// Exchange side: pages (host memory) -> flat device columns; returns the number of rows. The caller
// owns 'deviceBytes' (>= the sum of the page sizes): long strings are views into it.
inline int64_t prestoDeserialize(const std::vector<const void*>& pages, const std::vector<int64_t>& sizes,
                                 const std::vector<int32_t>& types, int32_t flags, void* deviceBytes,
                                 int64_t deviceBytesCapacity, vx355_out_column* columns, int64_t capacityRows) {
  int64_t rows = 0;
  check(vx355_presto_deserialize(pages.data(), sizes.data(), static_cast<int32_t>(pages.size()), types.data(),
                                 static_cast<int32_t>(types.size()), flags, deviceBytes, deviceBytesCapacity, columns,
                                 capacityRows, &rows));
  return rows;
}

// The table HashJoinBridge hands from build to probe (exec/HashJoinBridge.h:57,116).
class JoinTable {
 public:
  explicit JoinTable(vx355_join_table* t = nullptr) : t_(t) {}
  JoinTable(const JoinTable& o) : t_(o.t_) {
    if (t_) {
      vx355_join_table_retain(t_);
    }
  }
  JoinTable(JoinTable&& o) noexcept : t_(o.t_) { o.t_ = nullptr; }
  JoinTable& operator=(JoinTable o) {
    std::swap(t_, o.t_);
    return *this;
  }
  ~JoinTable() {
    if (t_) {
      vx355_join_table_release(t_);
    }
  }
  vx355_join_table* get() const { return t_; }
  vx355_join_table_stats stats() const {
    vx355_join_table_stats s{};
    check(vx355_join_table_get_stats(t_, &s));
    return s;
  }
  // HashProbe::pushdownDynamicFilters: what to push to the probe-side scan for key 'key'.
  vx355_key_filter keyFilter(int32_t key) const {
    vx355_key_filter f{};
    check(vx355_join_table_key_filter(t_, key, &f));
    return f;
  }

 private:
  vx355_join_table* t_;
};

// exec::HashBuild (exec/HashBuild.h): one per build Driver.
class HashBuild {
 public:
  HashBuild(std::vector<int32_t> keyChannels, std::vector<int32_t> keyTypes, std::vector<int32_t> dependentChannels,
            std::vector<int32_t> dependentTypes, vx355_join_type joinType = VX355_JOIN_INNER, bool nullAware = false,
            bool nullAsValue = false, bool dropDuplicates = false /* HashJoinNode::canDropDuplicates */)
      : keyChannels_(std::move(keyChannels)),
        keyTypes_(std::move(keyTypes)),
        dependentChannels_(std::move(dependentChannels)),
        dependentTypes_(std::move(dependentTypes)) {
    vx355_join_build_spec spec{};
    spec.num_keys = static_cast<int32_t>(keyChannels_.size());
    spec.key_cols = keyChannels_.data();
    spec.key_types = keyTypes_.data();
    spec.num_dependents = static_cast<int32_t>(dependentChannels_.size());
    spec.dependent_cols = dependentChannels_.data();
    spec.dependent_types = dependentTypes_.data();
    spec.join_type = joinType;
    spec.null_aware = nullAware ? 1 : 0;
    spec.null_as_value = nullAsValue ? 1 : 0;
    spec.drop_duplicates = dropDuplicates ? 1 : 0;
    check(vx355_join_build_create(&spec, &handle_));
  }
  HashBuild(const HashBuild&) = delete;
  HashBuild& operator=(const HashBuild&) = delete;
  ~HashBuild() { close(); }

  bool needsInput() const { return !finished_; }
  void addInput(const vx355_batch& input) { check(vx355_join_build_add_input(handle_, &input)); }
  int64_t addInputAsync(const vx355_batch& input) {   // see HashAggregation::addInputAsync
    int64_t ticket = 0;
    check(vx355_join_build_add_input_async(handle_, &input, &ticket));
    return ticket;
  }
  int64_t inFlight() const {
    int64_t submitted = 0, completed = 0;
    check(vx355_join_build_poll(handle_, &submitted, &completed));
    return submitted - completed;
  }
  void wait() { check(vx355_join_build_wait(handle_)); }
  // HashBuild::noMoreInput + finishHashBuild (HashBuild.cpp:819-993): called on the LAST of the
  // peer build operators with the others; returns the table the bridge publishes.
  JoinTable noMoreInput(const std::vector<HashBuild*>& peers = {}) {
    std::vector<vx355_join_build*> raw;
    for (auto* p : peers) {
      raw.push_back(p->handle_);
    }
    vx355_join_table* t = nullptr;
    check(vx355_join_build_finish(handle_, raw.data(), static_cast<int32_t>(raw.size()), &t));
    finished_ = true;
    for (auto* p : peers) {
      p->finished_ = true;
    }
    return JoinTable(t);
  }
  bool isFinished() const { return finished_; }
  void close() {
    if (handle_) {
      vx355_join_build_destroy(handle_);
      handle_ = nullptr;
    }
  }

 private:
  std::vector<int32_t> keyChannels_, keyTypes_, dependentChannels_, dependentTypes_;
  vx355_join_build* handle_ = nullptr;
  bool finished_ = false;
};

// exec::HashProbe (exec/HashProbe.h).
class HashProbe {
 public:
  HashProbe(const JoinTable& table, std::vector<int32_t> keyChannels, vx355_join_type joinType = VX355_JOIN_INNER,
            bool nullAware = false, bool nullAsValue = false)
      : keyChannels_(std::move(keyChannels)), joinType_(joinType) {
    vx355_join_probe_spec spec{};
    spec.num_keys = static_cast<int32_t>(keyChannels_.size());
    spec.key_cols = keyChannels_.data();
    spec.join_type = joinType;
    spec.null_aware = nullAware ? 1 : 0;
    spec.null_as_value = nullAsValue ? 1 : 0;
    check(vx355_join_probe_create(table.get(), &spec, &handle_));
  }
  HashProbe(const HashProbe&) = delete;
  HashProbe& operator=(const HashProbe&) = delete;
  ~HashProbe() { close(); }

  bool needsInput() const { return !noMoreInput_ && drained_; }  // HashProbe.h: one input batch at a time
  void addInput(const vx355_batch& input) {
    check(vx355_join_probe_add_input(handle_, &input));
    drained_ = false;
  }
  void noMoreInput() { noMoreInput_ = true; }
  // One output batch: mapping[i] = probe row, buildRows[i] = build row or -1, buildColumns gathered at
  // buildRows (extractColumns). Returns the row count; 0 = the current input is drained.
  int32_t getOutput(int32_t maxRows, int32_t* mapping, int32_t* buildRows, vx355_out_column* buildColumns = nullptr,
                    const int32_t* buildColumnIds = nullptr, int32_t numBuildColumns = 0, int32_t mem = VX355_MEM_HOST) {
    if (drained_) {
      return 0;
    }
    int32_t n = 0, finished = 0;
    check(vx355_join_probe_get_output(handle_, maxRows, mapping, buildRows, mem, buildColumns, buildColumnIds,
                                      numBuildColumns, &n, &finished));
    drained_ = finished != 0;
    return n;
  }
  // HashProbe::getBuildSideOutput (right / full / right semi), on the last prober after noMoreInput.
  int32_t getBuildSideOutput(int32_t maxRows, int32_t* buildRows, vx355_out_column* buildColumns = nullptr,
                             const int32_t* buildColumnIds = nullptr, int32_t numBuildColumns = 0,
                             int32_t mem = VX355_MEM_HOST) {
    if (buildSideDone_) {
      return 0;
    }
    int32_t n = 0, finished = 0;
    check(vx355_join_probe_get_build_side_output(handle_, maxRows, buildRows, mem, buildColumns, buildColumnIds,
                                                 numBuildColumns, &n, &finished));
    buildSideDone_ = finished != 0;
    return n;
  }
  bool isFinished() const { return noMoreInput_ && drained_; }
  void close() {
    if (handle_) {
      vx355_join_probe_destroy(handle_);
      handle_ = nullptr;
    }
  }

 private:
  std::vector<int32_t> keyChannels_;
  vx355_join_type joinType_;
  vx355_join_probe* handle_ = nullptr;
  bool noMoreInput_ = false;
  bool drained_ = true;
  bool buildSideDone_ = false;
};

// ---- multi-GPU (vx355.h "multi-GPU exchange"): one Communicator per GPU ---------------------------
class Communicator {
 public:
  // one process per GPU: the id comes from uniqueId() on rank 0 and travels out of band
  Communicator(const std::vector<char>& id, int32_t world, int32_t rank) {
    check(vx355_comm_create(id.data(), world, rank, &handle_));
  }
  Communicator(const Communicator&) = delete;
  Communicator& operator=(const Communicator&) = delete;
  ~Communicator() { vx355_comm_destroy(handle_); }
  static std::vector<char> uniqueId() {
    std::vector<char> id(VX355_COMM_ID_BYTES);
    check(vx355_comm_get_unique_id(id.data()));
    return id;
  }
  int32_t worldSize() const {  // as RCCL reports it
    int32_t world = 0;
    check(vx355_comm_info(handle_, &world, nullptr, nullptr));
    return world;
  }
  vx355_comm* get() const { return handle_; }

 private:
  vx355_comm* handle_ = nullptr;
};

// One PartitionedOutput(keys) -> Exchange edge of a repartitioned plan (vx355_exchange_*).
class Exchange {
 public:
  Exchange(Communicator& comm, std::vector<int32_t> columnTypes, std::vector<int32_t> keyChannels)
      : types_(std::move(columnTypes)), keys_(std::move(keyChannels)) {
    check(vx355_exchange_create(comm.get(), types_.data(), static_cast<int32_t>(types_.size()), keys_.data(),
                                static_cast<int32_t>(keys_.size()), &handle_));
  }
  Exchange(const Exchange&) = delete;
  Exchange& operator=(const Exchange&) = delete;
  ~Exchange() { vx355_exchange_destroy(handle_); }
  // PartitionedOutput::addInput: returns while the slices are on the links (two may be in flight)
  void send(const vx355_batch& input) { check(vx355_exchange_send(handle_, &input)); }
  // Exchange::getOutput: the rows that landed here, as device columns valid until the next receive
  vx355_batch receive(std::vector<vx355_column>& columns) {
    columns.resize(types_.size());
    int64_t rows = 0;
    check(vx355_exchange_receive(handle_, columns.data(), &rows));
    return vx355_batch{static_cast<int32_t>(rows), static_cast<int32_t>(columns.size()), columns.data()};
  }

 private:
  std::vector<int32_t> types_, keys_;
  vx355_exchange* handle_ = nullptr;
};

// BASELINE config 5 in one call (vx355_join_repartition): sink(chunk, received rows, probe handle)
// drains the probe of every chunk. F: void(int32_t, const vx355_batch&, vx355_join_probe*).
template <typename F>
JoinTable repartitionedJoin(Communicator& comm, const vx355_join_build_spec& buildSpec, const vx355_batch& buildRows,
                            const vx355_join_probe_spec& probeSpec, const vx355_batch& probeRows, int32_t chunks,
                            F&& sink) {
  struct Thunk {
    F* f;
    std::string error;
    static int call(void* arg, int32_t chunk, const vx355_batch* received, vx355_join_probe* probe) {
      auto* self = static_cast<Thunk*>(arg);
      try {
        (*self->f)(chunk, *received, probe);
        return VX355_OK;
      } catch (const std::exception& e) {  // no exception crosses the C frames
        self->error = e.what();
        return VX355_EINTERNAL;
      }
    }
  } thunk{&sink, {}};
  vx355_join_table* table = nullptr;
  const int status = vx355_join_repartition(comm.get(), &buildSpec, &buildRows, &probeSpec, &probeRows, chunks,
                                            &Thunk::call, &thunk, &table);
  if (!thunk.error.empty()) {
    throw RuntimeError(VX355_EINTERNAL, thunk.error);
  }
  check(status);
  return JoinTable(table);
}

// partial -> PrestoPages -> every rank -> final (vx355_agg_merge_partials): the returned handle has
// consumed all ranks' partial rows; drain it with vx355_agg_get_output, then vx355_agg_destroy.
inline vx355_agg* mergePartials(Communicator& comm, vx355_agg* partial, const vx355_agg_spec& finalSpec) {
  vx355_agg* fin = nullptr;
  check(vx355_agg_merge_partials(comm.get(), partial, &finalSpec, &fin));
  return fin;
}

}  // namespace vx355

#endif  // VX355_HPP_
