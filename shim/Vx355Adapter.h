// Velox-side adapter of libvx355: replaces exec::HashAggregation (this file) and exec::HashBuild /
// exec::HashProbe (Vx355JoinAdapter.h) in a Driver by operators that run on the MI355X through the C
// ABI of include/vx355.h. Built on the VELOX side (this
// repository has no Velox to compile against): add shim/ to a Velox build with
// shim/CMakeLists.txt and call facebook::velox::vx355::registerVx355() once per process, before
// the first Task starts.
//
// Extension point: exec::DriverFactory::registerAdapter (exec/Driver.h:789-847), the one cuDF uses
// (experimental/cudf/exec/ToCudf.cpp:277-295). Operator contract: exec/Operator.h:241-299.
#pragma once

#include <memory>
#include <vector>

#include "velox/exec/Driver.h"
#include "velox/exec/Operator.h"
#include "velox/vector/ComplexVector.h"
#include "velox/vector/DecodedVector.h"
#include "vx355.h"

namespace facebook::velox::vx355 {

/// Registers the adapter: every Driver created afterwards has its HashAggregation, HashBuild and
/// HashProbe operators replaced where libvx355 supports the plan (the create call succeeds);
/// everything else stays on the CPU operators. device: the GPU of this process (one process per GPU).
void registerVx355(int device = 0);

/// A RowVector reduced to what DecodedVector exposes per child: the vx355_batch of one addInput.
/// Keeps the DecodedVectors (and through them the input buffers) alive while the library reads.
class DecodedBatch {
 public:
  explicit DecodedBatch(const RowVector& input);
  const vx355_batch* get() const {
    return &batch_;
  }

 private:
  std::vector<DecodedVector> decoded_;
  std::vector<vx355_column> columns_;
  std::vector<std::vector<uint64_t>> flippedNulls_;  // (unused: Velox nulls are 1 = valid, like vx355's)
  vx355_batch batch_{};
};

/// vx355_out_column descriptors over the children of a result RowVector Velox allocated.
class OutColumns {
 public:
  explicit OutColumns(RowVector& result);
  vx355_out_column* data() {
    return columns_.data();
  }
  int32_t size() const {
    return static_cast<int32_t>(columns_.size());
  }

 private:
  std::vector<vx355_out_column> columns_;
};

/// VARCHAR / VARBINARY columns the library filled: views of more than 12 bytes point into a buffer the
/// handle only keeps until its next output call - copy those strings into the vector's own buffers.
void ownStrings(const VectorPtr& column, vector_size_t numRows);

/// exec::HashAggregation on the GPU (exec/HashAggregation.h). Input batches are queued through the
/// asynchronous boundary (vx355_agg_add_input_async): the Driver thread does not wait for staging
/// copies, transfers and kernels; isBlocked() bounds the batches in flight.
class Vx355HashAggregation : public exec::Operator {
 public:
  Vx355HashAggregation(
      int32_t operatorId,
      exec::DriverCtx* driverCtx,
      const std::shared_ptr<const core::AggregationNode>& node,
      vx355_agg* handle);
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

  vx355_agg* handle_;
  const bool isPartialOutput_;
  const bool isGlobal_;
  const int64_t maxPartialMemory_;
  bool flushing_{false};
  bool finished_{false};
  // batches handed to the library and not yet reported complete: (ticket, input, decoded view)
  struct InFlight {
    int64_t ticket;
    RowVectorPtr input;
    std::unique_ptr<DecodedBatch> decoded;
  };
  std::vector<InFlight> inFlight_;
};

}  // namespace facebook::velox::vx355
