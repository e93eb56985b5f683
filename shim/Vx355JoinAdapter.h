// Velox-side adapter of libvx355, join half: replaces exec::HashBuild / exec::HashProbe of a
// HashJoinNode by operators that run on the MI355X through the C ABI of include/vx355.h
// (INTEGRATION.md section 3). Built on the VELOX side together with Vx355Adapter.cpp; registered by
// the same registerVx355() call.
//
// The table travels from the build pipeline to the probe pipeline through a small rendezvous of its
// own (Vx355JoinTables) instead of exec::HashJoinBridge, whose payload is a BaseHashTable
// (exec/HashJoinBridge.h:57): the last build Driver publishes the vx355_join_table*, probe Drivers
// that come earlier get a ContinueFuture (exec/HashProbe.cpp:527 does the same with the bridge).
#pragma once

#include <atomic>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "Vx355Adapter.h"
#include "velox/common/future/VeloxPromise.h"
#include "velox/core/PlanNode.h"
#include "velox/type/Filter.h"

namespace facebook::velox::vx355 {

/// (task id, split group, plan node id) -> table, for the Drivers of this process.
class Vx355JoinTables {
 public:
  using Key = std::tuple<std::string, uint32_t, core::PlanNodeId>;
  static Vx355JoinTables& instance();

  /// A probe operator exists for 'key' (constructor) / is gone (close). The entry and the
  /// rendezvous' reference on the table go away with the last one.
  void addProbe(const Key& key);
  void removeProbe(const Key& key);
  /// The table, or nullptr and a future that completes when it is published.
  vx355_join_table* tableOrFuture(const Key& key, ContinueFuture* future);
  /// Build side (last peer): hands over one reference on 'table'. Without a registered probe (the task
  /// is being torn down: every probe operator is already closed) the reference is dropped at once.
  void publish(const Key& key, vx355_join_table* table);
  /// Whether the library takes the join of 'key': decided once (by 'decide', a trial
  /// vx355_join_build_create) for all Drivers of both pipelines, so that they decide alike.
  bool accepted(const Key& key, const std::function<bool()>& decide);
  /// true for the first probe operator of 'key' that asks about join key 'channel': it makes the dynamic
  /// filter, its peers find it made (HashProbe::dynamicFiltersProducedOnChannels_, exec/HashProbe.cpp:419-446).
  bool firstToFilter(const Key& key, int32_t channel);

 private:
  struct Entry {
    vx355_join_table* table{nullptr};
    int32_t probes{0};
    int32_t accepted{-1};  // -1 = not decided yet
    std::vector<ContinuePromise> promises;
    std::vector<int32_t> filteredChannels;
  };
  std::mutex mutex_;
  std::map<Key, Entry> entries_;
};

/// What both operators need of a HashJoinNode, resolved against the types of its two sources.
struct JoinPlan {
  std::vector<int32_t> probeKeys, buildKeys, buildKeyTypes;
  // build-side columns the join emits: their channels in the build input (= the table's dependent
  // columns, in this order), their types, and where each one goes in the output row
  std::vector<int32_t> dependentChannels, dependentTypes, dependentOutputs;
  // probe-side columns the join emits: (probe channel, output channel)
  std::vector<std::pair<int32_t, int32_t>> probeOutputs;
  int32_t matchOutput{-1};  // semi project joins: the BOOLEAN 'match' column
  vx355_join_type type{VX355_JOIN_INNER};
  bool nullAware{false}, nullAsValue{false}, dropDuplicates{false};
  // FilterProject -> HashProbe fusion: the terms of the FilterNode in front of the probe (empty: none)
  std::vector<vx355_filter_term> inputFilter;
};

/// Join kinds whose unmatched probe rows emit nothing: the only ones a filter in front of the probe can
/// be folded into (vx355_join_probe_set_input_filter).
bool fusesInputFilter(const JoinPlan& plan);

/// false: a join the library does not take (extra filter outside INTEGRATION.md's class, key or
/// payload types beyond the scalar kinds): the CPU operators stay.
bool toJoinPlan(const core::HashJoinNode& node, JoinPlan* out);

/// exec::HashBuild on the GPU (exec/HashBuild.h): one per build Driver.
class Vx355HashBuild : public exec::Operator {
 public:
  Vx355HashBuild(
      int32_t operatorId,
      exec::DriverCtx* driverCtx,
      const std::shared_ptr<const core::HashJoinNode>& node,
      const JoinPlan& plan,
      vx355_join_build* handle);
  ~Vx355HashBuild() override;

  bool needsInput() const override {
    return !noMoreInput_;
  }
  void addInput(RowVectorPtr input) override;
  void noMoreInput() override;
  RowVectorPtr getOutput() override {
    return nullptr;
  }
  exec::BlockingReason isBlocked(ContinueFuture* future) override;
  bool isFinished() override {
    return finished_.load();
  }
  void close() override;
  vx355_join_build* handle() const {
    return handle_;
  }
  /// (the last peer, from its own Driver thread, once the table is published)
  void markFinished() {
    finished_.store(true);
  }

 private:
  vx355_join_build* handle_;
  Vx355JoinTables::Key key_;
  ContinueFuture future_{ContinueFuture::makeEmpty()};
  std::atomic<bool> finished_{false};
};

/// exec::HashProbe on the GPU (exec/HashProbe.h).
class Vx355HashProbe : public exec::Operator {
 public:
  Vx355HashProbe(
      int32_t operatorId,
      exec::DriverCtx* driverCtx,
      const std::shared_ptr<const core::HashJoinNode>& node,
      JoinPlan plan);
  ~Vx355HashProbe() override;

  bool needsInput() const override;
  void addInput(RowVectorPtr input) override;
  void noMoreInput() override;
  RowVectorPtr getOutput() override;
  exec::BlockingReason isBlocked(ContinueFuture* future) override;
  bool isFinished() override;
  void close() override;

 private:
  RowVectorPtr fillOutput(int32_t numRows, const BufferPtr& mapping, const int32_t* buildRows,
                          std::vector<VectorPtr>& buildColumns, bool buildSide);
  bool emitsBuildSide() const;
  /// HashProbe::pushdownDynamicFilters (exec/HashProbe.cpp:408-457): one filter per join key whose column an
  /// upstream operator accepts filters on (Driver::pushdownFilters walks there through identity projections),
  /// made from the finished table: value list -> common::createBigintValues, or the table's Bloom blocks.
  void pushdownDynamicFilters();
  void recordStats();

  const JoinPlan plan_;
  Vx355JoinTables::Key key_;
  vx355_join_table* table_{nullptr};
  vx355_join_probe* handle_{nullptr};
  ContinueFuture future_{ContinueFuture::makeEmpty()};
  // (Operator::input_ keeps the batch until its output is drained, as in exec::HashProbe)
  std::unique_ptr<DecodedBatch> decoded_;
  bool inputDrained_{true};
  bool lastProber_{false}, buildSideDone_{false}, finished_{false};
  bool statsRecorded_{false};
  // The page of output the library's worker is filling (vx355_join_probe_get_output_async): queued by
  // isBlocked() behind the batch, handed to the Driver by getOutput() when the callback has fired.
  struct Page {
    BufferPtr mapping;
    std::vector<int32_t> buildRows;
    std::vector<VectorPtr> buildColumns;
    std::vector<vx355_out_column> out;
    std::vector<int32_t> ids;
    bool buildSide{false};
    int64_t ticket{0};
    std::mutex mutex;   // (a promise per isBlocked() call, as in Vx355HashAggregation)
    std::vector<ContinuePromise> promises;
    std::atomic<bool> done{false};
  };
  std::unique_ptr<Page> page_;
  bool wantsBuildSide() const {
    return inputDrained_ && noMoreInput_ && lastProber_ && !buildSideDone_;
  }
  void startPage(bool buildSide);
  static void onPageDone(void* arg, int status, int32_t numRows, int32_t finished);
};

/// Called by the adapter of Vx355Adapter.cpp for every Driver: replaces the HashBuild / HashProbe
/// operators of joins the library takes. Both pipelines of a join decide alike (the decision only
/// depends on the plan node).
bool adaptJoins(const exec::DriverFactory& factory, exec::Driver& driver);

}  // namespace facebook::velox::vx355
