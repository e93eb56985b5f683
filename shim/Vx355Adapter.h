// Velox-side adapter of libvx355: replaces exec::HashAggregation (this file; with the FilterProject in
// front of it when its expressions are in the library's class) and exec::HashBuild / exec::HashProbe
// (Vx355JoinAdapter.h) in a Driver by operators that run on the MI355X through the C ABI of
// include/vx355.h. Built on the VELOX side: add shim/ to a Velox build with shim/CMakeLists.txt and
// call facebook::velox::vx355::registerVx355() once per process, before the first Task starts.
// In this repository the sources are compiled and run against tests/velox_api_stub (declarations with
// the reference's signatures): tests/test_shim.py.
//
// Extension point: exec::DriverFactory::registerAdapter (exec/Driver.h:789-847), the one cuDF uses
// (experimental/cudf/exec/ToCudf.cpp:277-295). Operator contract: exec/Operator.h:241-299.
#pragma once

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "velox/common/future/VeloxPromise.h"
#include "velox/core/PlanNode.h"
#include "velox/exec/Driver.h"
#include "velox/exec/HashTable.h"
#include "velox/exec/Operator.h"
#include "velox/vector/ComplexVector.h"
#include "velox/vector/DecodedVector.h"
#include "vx355.h"

namespace facebook::velox::vx355 {

/// Registers the adapter: every Driver created afterwards has its HashAggregation, HashBuild and
/// HashProbe operators replaced where libvx355 supports the plan (the create call succeeds);
/// everything else stays on the CPU operators. device: the GPU of this process (one process per GPU).
/// memoryLimitBytes: the share of the GPU's HBM the embedding gives to the operators' tables, staged input
/// and scratch (0 = all of it); past it an operator fails with VELOX_MEM_POOL_CAP_EXCEEDED - the library does
/// not spill (vx355_set_memory_limit, include/vx355.h).
void registerVx355(int device = 0, int64_t memoryLimitBytes = 0);

/// The adapter function itself (what registerVx355 registers); exposed for tests.
bool adaptDriver(const exec::DriverFactory& factory, exec::Driver& driver);

/// Where the channels of a RowType land in a vx355_batch. A scalar channel is one vx355_column; a
/// ROW(scalar, ...) channel - the intermediate type of avg, ROW(DOUBLE sum, BIGINT count)
/// (functions/lib/aggregates/AverageAggregateBase.h:66-260) - is flattened into one column per field,
/// later channels move up. Any other type occupies one column nobody may reference.
struct ColumnLayout {
  std::vector<int32_t> first;  // first vx355 column of channel c
  std::vector<int32_t> count;  // its number of vx355 columns
  int32_t numColumns{0};
  explicit ColumnLayout(const RowTypePtr& type);
  ColumnLayout() = default;
};

/// A constant argument of an aggregate (exec/AggregateInfo.cpp:62-69 hands those to the function as
/// constant vectors): travels as an extra VX355_CONSTANT column behind the batch's own columns.
struct ConstantColumn {
  int32_t typeKind{VX355_BIGINT};
  bool isNull{false};
  alignas(16) char value[16]{};  // the value in its column representation (StringView for strings)
  std::string longString;        // bytes of a string constant longer than 12 bytes
};

/// A RowVector reduced to what DecodedVector exposes per child: the vx355_batch of one addInput.
/// Keeps the DecodedVectors (and through them the input buffers) alive while the library reads.
class DecodedBatch {
 public:
  DecodedBatch(const RowVector& input, const ColumnLayout& layout, const std::vector<ConstantColumn>& constants);
  /// all channels scalar, no constants
  explicit DecodedBatch(const RowVector& input);
  const vx355_batch* get() const {
    return &batch_;
  }

 private:
  void addChild(const BaseVector& child, vector_size_t numRows);
  void addStruct(const VectorPtr& child, int32_t numFields, vector_size_t numRows);
  std::vector<std::unique_ptr<DecodedVector>> decoded_;
  std::vector<VectorPtr> flattened_;                  // structs that arrived wrapped
  std::vector<std::vector<uint64_t>> mergedNulls_;    // field nulls AND struct nulls
  std::vector<vx355_column> columns_;
  vx355_batch batch_{};
};

/// vx355_out_column descriptors over the children of a result RowVector Velox allocated. A ROW child
/// (partial avg out) contributes one descriptor per field; finish() derives the struct's own nulls.
class OutColumns {
 public:
  explicit OutColumns(RowVector& result);
  vx355_out_column* data() {
    return columns_.data();
  }
  int32_t size() const {
    return static_cast<int32_t>(columns_.size());
  }
  /// After the library filled 'numRows' rows: a struct is null where its first field is
  /// (AverageAggregateBase::extractAccumulators sets the row null for a group without input).
  void finish(vector_size_t numRows);

 private:
  RowVector& result_;
  std::vector<vx355_out_column> columns_;
};

/// VARCHAR / VARBINARY columns the library filled: views of more than 12 bytes point into a buffer the
/// handle only keeps until its next output call - copy those strings into the vector's own buffers.
void ownStrings(const VectorPtr& column, vector_size_t numRows);

/// What one channel of the aggregation's input is in terms of the batches the operator receives:
/// a column of the batch (the first one of a flattened struct), VX355_PROJECTION_COL_BASE + j for
/// projection j of a fused FilterProject, or -1 when the channel cannot be consumed.
struct InputBinding {
  int32_t column{-1};
  TypePtr type;
};

/// FilterProject -> HashAggregation fusion (vx355_agg_set_fused_input): the filter and the projections
/// of the FilterNode / ProjectNode in front of the aggregation in the library's expression class
/// (include/vx355.h "FilterProject for the TPC-H Q1 / Q3 expression class").
struct FusedInput {
  RowTypePtr scanType;  // the FilterProject's input type = what the fused operator receives
  std::vector<vx355_filter_term> terms;
  std::vector<vx355_projection> projections;
  std::vector<InputBinding> bindings;  // per channel of the aggregation's input (the project's output)
};

/// false: the expressions are outside the class (the FilterProject stays a CPU operator).
/// 'filter' and 'project' may each be null (not both).
bool toFusedInput(const core::FilterNode* filter, const core::ProjectNode* project, FusedInput* out);

/// The filter of a FilterNode alone as vx355_filter_terms over the columns of its input (scalar channels:
/// column = channel): what vx355_join_probe_set_input_filter takes (Vx355JoinAdapter.cpp).
bool toFilterTerms(const core::FilterNode& filter, std::vector<vx355_filter_term>* out);

/// vx355_agg_spec of an AggregationNode: keys, aggregates, step, ignoreNullKeys
/// (core/PlanNode.h:1120-1370). Returns false when the plan is outside what the library takes - the
/// CPU operator then stays in place, like cuDF's adapter does (ToCudf.cpp:230-242).
struct AggSpec {
  std::vector<int32_t> keyCols, keyTypes;
  std::vector<vx355_agg_fn> fns;
  std::vector<ConstantColumn> constants;  // columns numColumns, numColumns + 1, ... of every batch
  ColumnLayout layout;                    // of the batches the operator receives
  vx355_agg_spec c{};
};
/// bindings: one per channel of node.sources()[0]->outputType() (identity over 'layout' when the
/// operator is not fused); layout: of the RowVectors the operator will receive.
bool toAggSpec(const core::AggregationNode& node, const std::vector<InputBinding>& bindings, const ColumnLayout& layout,
               AggSpec* out);
/// The unfused form: bindings straight from the node's input type.
bool toAggSpec(const core::AggregationNode& node, AggSpec* out);

/// exec::HashAggregation on the GPU (exec/HashAggregation.h). Input batches are queued through the
/// asynchronous boundary (vx355_agg_add_input_async): the Driver thread does not wait for staging
/// copies, transfers and kernels; isBlocked() bounds the batches in flight. noMoreInput and the output
/// pages are queued too (vx355_agg_no_more_input_async / vx355_agg_get_output_async): isBlocked() hands the
/// Driver a future that the library's worker fulfils when the page is in the result vector.
/// The gpu.* runtime stats of one operator: busy nanoseconds, bytes over PCIe in both
/// directions, input bytes the kernels read, bytes written (output + the operator's table), launches.
void recordGpuStats(exec::Operator& op, const vx355_gpu_stats& gpu, int64_t tableBytes);

class Vx355HashAggregation : public exec::Operator {
 public:
  Vx355HashAggregation(
      int32_t operatorId,
      exec::DriverCtx* driverCtx,
      const std::shared_ptr<const core::AggregationNode>& node,
      vx355_agg* handle,
      ColumnLayout layout,
      std::vector<ConstantColumn> constants);
  ~Vx355HashAggregation() override;

  bool needsInput() const override;
  void addInput(RowVectorPtr input) override;
  void noMoreInput() override;
  RowVectorPtr getOutput() override;
  exec::BlockingReason isBlocked(ContinueFuture* future) override;
  bool isFinished() override;
  void close() override;
  bool canReclaim() const override {
    return false;  // the state lives in HBM; partial aggregations shed it with vx355_agg_flush
  }

 private:
  static void check(int status);
  void releaseCompleted();
  bool partialFull();
  /// hashtable.* (exec/HashTable.h:155-159) and gpu.* runtime stats, once, when the operator is done.
  void recordStats();
  bool wantsOutput() const {
    return !finished_ && (noMoreInput_ || flushing_);
  }
  void startPage();
  static void onPageDone(void* arg, int status, int32_t numRows, int32_t finished);

  vx355_agg* handle_;
  const ColumnLayout layout_;
  const std::vector<ConstantColumn> constants_;
  const bool isPartialOutput_;
  const bool isGlobal_;
  const int64_t maxPartialMemory_;
  bool flushing_{false};
  bool finished_{false};
  int64_t completedAtLastCheck_{0};
  bool statsRecorded_{false};
  // batches handed to the library and not yet reported complete: (ticket, input, decoded view)
  struct InFlight {
    int64_t ticket;
    RowVectorPtr input;
    std::unique_ptr<DecodedBatch> decoded;
  };
  std::vector<InFlight> inFlight_;
  // the output page being filled by the library's worker (one at a time: the next one is queued when
  // this one has been handed to the Driver)
  struct Page {
    RowVectorPtr result;
    std::unique_ptr<OutColumns> out;
    int64_t ticket{0};
    // (a promise per isBlocked() call: a folly promise hands out its future once; all are fulfilled together)
    std::mutex mutex;
    std::vector<ContinuePromise> promises;
    std::atomic<bool> done{false};
  };
  std::unique_ptr<Page> page_;
};

}  // namespace facebook::velox::vx355
