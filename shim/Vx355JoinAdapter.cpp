// See Vx355JoinAdapter.h. Compiled on the Velox side only.
#include "Vx355JoinAdapter.h"

#include <thread>

#include <algorithm>
#include <chrono>

#include "velox/core/QueryConfig.h"
#include "velox/exec/FilterProject.h"
#include "velox/exec/HashBuild.h"
#include "velox/exec/HashProbe.h"
#include "velox/exec/OperatorUtils.h"
#include "velox/exec/Task.h"
#include "velox/common/memory/MemoryArbitrator.h"
#include "velox/vector/FlatVector.h"

namespace facebook::velox::vx355 {

namespace {

void check(int status) {
  if (status == VX355_OK) {
    return;
  }
  if (status == VX355_ENOMEM) {
    // The library's share of HBM (registerVx355's memoryLimitBytes / the GPU itself) is exhausted and it
    // does not spill: the failure an operator reports when its MemoryPool cannot grow and nothing can be
    // reclaimed (canReclaim() == false). The handle stays destroyable; the Task fails cleanly.
    VELOX_MEM_POOL_CAP_EXCEEDED("{}", vx355_last_error());
  }
  if (status == VX355_EUSER) {
    VELOX_USER_FAIL("{}", vx355_last_error());
  }
  VELOX_FAIL("{}", vx355_last_error());
}

// TypeKind values equal vx355_type_kind (type/TypeKind.h:41-52); DATE is INTEGER.
bool scalarKind(TypeKind kind) {
  switch (kind) {
    case TypeKind::BOOLEAN:
    case TypeKind::TINYINT:
    case TypeKind::SMALLINT:
    case TypeKind::INTEGER:
    case TypeKind::BIGINT:
    case TypeKind::REAL:
    case TypeKind::DOUBLE:
    case TypeKind::VARCHAR:
    case TypeKind::VARBINARY:
    case TypeKind::TIMESTAMP:
      return true;
    default:
      return false;
  }
}

Vx355JoinTables::Key keyOf(exec::DriverCtx* ctx, const core::PlanNodeId& id) {
  return {ctx->task->taskId(), ctx->splitGroupId, id};
}

std::shared_ptr<const core::HashJoinNode> joinNodeOf(const exec::DriverFactory& factory, const core::PlanNodeId& id) {
  for (const auto& node : factory.planNodes) {
    if (node->id() == id) {
      return std::dynamic_pointer_cast<const core::HashJoinNode>(node);
    }
  }
  // the build pipeline ends in the join node: it is that pipeline's consumer
  if (factory.consumerNode && factory.consumerNode->id() == id) {
    return std::dynamic_pointer_cast<const core::HashJoinNode>(factory.consumerNode);
  }
  return nullptr;
}

}  // namespace

// ---- rendezvous --------------------------------------------------------------------------------

Vx355JoinTables& Vx355JoinTables::instance() {
  static Vx355JoinTables tables;
  return tables;
}

void Vx355JoinTables::addProbe(const Key& key) {
  std::lock_guard<std::mutex> l(mutex_);
  ++entries_[key].probes;
}

void Vx355JoinTables::removeProbe(const Key& key) {
  vx355_join_table* release = nullptr;
  {
    std::lock_guard<std::mutex> l(mutex_);
    auto it = entries_.find(key);
    if (it == entries_.end() || --it->second.probes > 0) {
      return;
    }
    release = it->second.table;
    entries_.erase(it);
  }
  if (release != nullptr) {
    vx355_join_table_release(release);
  }
}

vx355_join_table* Vx355JoinTables::tableOrFuture(const Key& key, ContinueFuture* future) {
  std::lock_guard<std::mutex> l(mutex_);
  auto& entry = entries_[key];
  if (entry.table != nullptr) {
    return entry.table;
  }
  entry.promises.emplace_back("Vx355JoinTables::tableOrFuture");
  *future = entry.promises.back().getSemiFuture();
  return nullptr;
}

void Vx355JoinTables::publish(const Key& key, vx355_join_table* table) {
  std::vector<ContinuePromise> promises;
  bool orphan = false;
  {
    std::lock_guard<std::mutex> l(mutex_);
    auto it = entries_.find(key);
    if (it == entries_.end() || it->second.probes <= 0) {
      // every probe operator is gone already (the task aborted between their close() and this
      // build's finish): nobody will ever take the table
      orphan = true;
      if (it != entries_.end()) {
        promises = std::move(it->second.promises);
        entries_.erase(it);
      }
    } else {
      it->second.table = table;
      promises = std::move(it->second.promises);
    }
  }
  if (orphan && table != nullptr) {
    vx355_join_table_release(table);
  }
  for (auto& promise : promises) {
    promise.setValue();
  }
}

bool Vx355JoinTables::firstToFilter(const Key& key, int32_t channel) {
  std::lock_guard<std::mutex> lock(mutex_);
  auto& channels = entries_[key].filteredChannels;
  if (std::find(channels.begin(), channels.end(), channel) != channels.end()) {
    return false;
  }
  channels.push_back(channel);
  return true;
}

bool Vx355JoinTables::accepted(const Key& key, const std::function<bool()>& decide) {
  std::lock_guard<std::mutex> l(mutex_);
  auto it = entries_.find(key);
  if (it != entries_.end() && it->second.accepted == 1) {
    return true;
  }
  if (!decide()) {
    return false;  // a refused join keeps no entry (the create call fails in its argument checks: cheap to repeat)
  }
  entries_[key].accepted = 1;  // lives until the join's last probe operator closes (removeProbe)
  return true;
}

// ---- plan --------------------------------------------------------------------------------------

bool toJoinPlan(const core::HashJoinNode& node, JoinPlan* out) {
  if (node.filter() != nullptr) {
    // conjunctions of up to four comparisons could be handed to
    // vx355_join_probe_set_filter; this adapter leaves joins with a filter on the CPU
    return false;
  }
  const auto& probeType = node.sources()[0]->outputType();
  const auto& buildType = node.sources()[1]->outputType();
  for (size_t i = 0; i < node.leftKeys().size(); ++i) {
    const auto probeChannel = exec::exprToChannel(node.leftKeys()[i].get(), probeType);
    const auto buildChannel = exec::exprToChannel(node.rightKeys()[i].get(), buildType);
    if (probeChannel == kConstantChannel || buildChannel == kConstantChannel ||
        !scalarKind(buildType->childAt(buildChannel)->kind()) ||
        probeType->childAt(probeChannel)->kind() != buildType->childAt(buildChannel)->kind()) {
      return false;
    }
    out->probeKeys.push_back(static_cast<int32_t>(probeChannel));
    out->buildKeys.push_back(static_cast<int32_t>(buildChannel));
    out->buildKeyTypes.push_back(static_cast<int32_t>(buildType->childAt(buildChannel)->kind()));
  }
  out->type = static_cast<vx355_join_type>(node.joinType());  // same numeric values (core/PlanNode.h:3078-3150)
  out->nullAware = node.isNullAware();
  out->nullAsValue = node.isNullAsValue();
  out->dropDuplicates = node.canDropDuplicates();
  const auto& outputType = node.outputType();
  const bool semiProject = node.isLeftSemiProjectJoin() || node.isRightSemiProjectJoin();
  for (uint32_t i = 0; i < outputType->size(); ++i) {
    if (semiProject && i + 1 == outputType->size()) {
      out->matchOutput = static_cast<int32_t>(i);  // the 'match' column comes last (core/PlanNode.h:3352-3366)
      continue;
    }
    const auto& name = outputType->nameOf(i);
    if (auto probeChannel = probeType->getChildIdxIfExists(name)) {
      out->probeOutputs.emplace_back(static_cast<int32_t>(*probeChannel), static_cast<int32_t>(i));
    } else if (auto buildChannel = buildType->getChildIdxIfExists(name)) {
      if (!scalarKind(buildType->childAt(*buildChannel)->kind())) {
        return false;
      }
      out->dependentChannels.push_back(static_cast<int32_t>(*buildChannel));
      out->dependentTypes.push_back(static_cast<int32_t>(buildType->childAt(*buildChannel)->kind()));
      out->dependentOutputs.push_back(static_cast<int32_t>(i));
    } else {
      return false;
    }
  }
  return true;
}

bool fusesInputFilter(const JoinPlan& plan) {
  if (plan.nullAware) {
    return false;
  }
  switch (plan.type) {
    case VX355_JOIN_INNER:
    case VX355_JOIN_RIGHT:
    case VX355_JOIN_LEFT_SEMI_FILTER:
    case VX355_JOIN_COUNTING_LEFT_SEMI_FILTER:
    case VX355_JOIN_RIGHT_SEMI_FILTER:
    case VX355_JOIN_RIGHT_SEMI_PROJECT:
    case VX355_JOIN_RIGHT_ANTI:
      return true;
    default:
      return false;  // LEFT / FULL / LEFT_SEMI_PROJECT / ANTI emit probe rows that found nothing
  }
}

// ---- build -------------------------------------------------------------------------------------

Vx355HashBuild::Vx355HashBuild(
    int32_t operatorId,
    exec::DriverCtx* driverCtx,
    const std::shared_ptr<const core::HashJoinNode>& node,
    const JoinPlan& /*plan*/,
    vx355_join_build* handle)
    : Operator(driverCtx, nullptr, operatorId, node->id(), "Vx355HashBuild"),
      handle_(handle),
      key_(keyOf(driverCtx, node->id())) {}

Vx355HashBuild::~Vx355HashBuild() {
  if (handle_ != nullptr) {
    vx355_join_build_destroy(handle_);
  }
}

void Vx355HashBuild::addInput(RowVectorPtr input) {
  DecodedBatch batch(*input);
  check(vx355_join_build_add_input(handle_, batch.get()));  // the rows are in HBM when this returns
}

void Vx355HashBuild::noMoreInput() {
  Operator::noMoreInput();
  // HashBuild::finishHashBuild (exec/HashBuild.cpp:819-993): the last of the peer build Drivers
  // merges all of them into one table and publishes it.
  std::vector<ContinuePromise> promises;
  std::vector<std::shared_ptr<exec::Driver>> peers;
  if (!operatorCtx_->task()->allPeersFinished(planNodeId(), operatorCtx_->driver(), &future_, promises, peers)) {
    return;  // isBlocked() hands future_ to the Driver; the last peer completes it
  }
  std::vector<vx355_join_build*> others;
  for (auto& peer : peers) {
    auto* build = dynamic_cast<Vx355HashBuild*>(peer->findOperator(planNodeId()));
    VELOX_CHECK_NOT_NULL(build);
    others.push_back(build->handle());
  }
  vx355_join_table* table = nullptr;
  const auto buildStart = std::chrono::steady_clock::now();
  const int status =
      vx355_join_build_finish(handle_, others.data(), static_cast<int32_t>(others.size()), &table);
  // "the last peer is responsible for the promises' fulfillment even in case of an exception"
  // (exec/Task.h:585-587)
  for (auto& promise : promises) {
    promise.setValue();
  }
  check(status);
  {
    // HashBuild::addRuntimeStats (exec/HashBuild.cpp:953-957,1103-1145): the table's numbers under the
    // reference's names, the time of the build, and what the build cost the GPU
    const auto buildNanos = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - buildStart).count();
    vx355_join_table_stats stats{};
    check(vx355_join_table_get_stats(table, &stats));
    addRuntimeStat(exec::BaseHashTable::kCapacity, RuntimeCounter(stats.capacity));
    addRuntimeStat(exec::BaseHashTable::kNumRehashes, RuntimeCounter(0));  // built once from all rows: never rehashed
    addRuntimeStat(exec::BaseHashTable::kNumDistinct, RuntimeCounter(stats.num_distinct));
    addRuntimeStat(exec::BaseHashTable::kHashMode, RuntimeCounter(stats.hash_mode));
    addRuntimeStat(exec::BaseHashTable::kBuildWallNanos, RuntimeCounter(buildNanos, RuntimeCounter::Unit::kNanos));
    vx355_gpu_stats gpu{};
    check(vx355_join_build_get_gpu_stats(handle_, &gpu));
    recordGpuStats(*this, gpu, stats.capacity * 16);
  }
  Vx355JoinTables::instance().publish(key_, table);
  for (auto& peer : peers) {
    static_cast<Vx355HashBuild*>(peer->findOperator(planNodeId()))->markFinished();
  }
  finished_ = true;
}

exec::BlockingReason Vx355HashBuild::isBlocked(ContinueFuture* future) {
  if (!future_.valid()) {
    return exec::BlockingReason::kNotBlocked;
  }
  *future = std::move(future_);
  finished_ = true;  // woken up by the last peer: nothing left to do here
  return exec::BlockingReason::kWaitForJoinBuild;
}

void Vx355HashBuild::close() {
  if (handle_ != nullptr) {
    vx355_join_build_destroy(handle_);
    handle_ = nullptr;
  }
  Operator::close();
}

// ---- probe -------------------------------------------------------------------------------------

Vx355HashProbe::Vx355HashProbe(
    int32_t operatorId,
    exec::DriverCtx* driverCtx,
    const std::shared_ptr<const core::HashJoinNode>& node,
    JoinPlan plan)
    : Operator(driverCtx, node->outputType(), operatorId, node->id(), "Vx355HashProbe"),
      plan_(std::move(plan)),
      key_(keyOf(driverCtx, node->id())) {
  Vx355JoinTables::instance().addProbe(key_);
}

Vx355HashProbe::~Vx355HashProbe() {
  if (handle_ != nullptr) {
    vx355_join_probe_destroy(handle_);
  }
}

bool Vx355HashProbe::emitsBuildSide() const {
  return plan_.type == VX355_JOIN_RIGHT || plan_.type == VX355_JOIN_FULL || plan_.type == VX355_JOIN_RIGHT_SEMI_FILTER ||
      plan_.type == VX355_JOIN_RIGHT_SEMI_PROJECT || plan_.type == VX355_JOIN_RIGHT_ANTI;
}

exec::BlockingReason Vx355HashProbe::isBlocked(ContinueFuture* future) {
  if (handle_ != nullptr) {
    // a batch is being probed, or the unmatched build rows are due: the page is queued behind it and the
    // Driver leaves the thread until the library's worker has filled it
    if (!finished_ && (!inputDrained_ || wantsBuildSide())) {
      if (page_ == nullptr) {
        startPage(wantsBuildSide());
      }
      std::lock_guard<std::mutex> lock(page_->mutex);
      if (!page_->done.load(std::memory_order_acquire)) {
        page_->promises.emplace_back("Vx355HashProbe::getOutput");
        *future = page_->promises.back().getSemiFuture();
        return exec::BlockingReason::kWaitForConnector;
      }
    }
    return exec::BlockingReason::kNotBlocked;
  }
  table_ = Vx355JoinTables::instance().tableOrFuture(key_, future);  // exec/HashProbe.cpp:527
  if (table_ == nullptr) {
    return exec::BlockingReason::kWaitForJoinBuild;
  }
  vx355_join_probe_spec spec{};
  spec.num_keys = static_cast<int32_t>(plan_.probeKeys.size());
  spec.key_cols = plan_.probeKeys.data();
  spec.join_type = plan_.type;
  spec.null_aware = plan_.nullAware ? 1 : 0;
  spec.null_as_value = plan_.nullAsValue ? 1 : 0;
  check(vx355_join_probe_create(table_, &spec, &handle_));  // takes its own reference on the table
  if (!plan_.inputFilter.empty()) {
    // the FilterProject that stood in front of this operator (adaptJoins checked kind and terms)
    check(vx355_join_probe_set_input_filter(
        handle_, plan_.inputFilter.data(), static_cast<int32_t>(plan_.inputFilter.size())));
  }
  check(vx355_join_probe_set_output_batch_bytes(
      handle_, static_cast<int64_t>(operatorCtx_->driverCtx()->queryConfig().preferredOutputBatchBytes())));
  // exec/HashProbe.cpp:578-596: the join kinds whose unmatched probe rows come out of nothing, a table
  // that is not in generic hash mode (its keys have value statistics) and is not empty
  vx355_join_table_stats stats{};
  check(vx355_join_table_get_stats(table_, &stats));
  const bool filterable = plan_.type == VX355_JOIN_INNER || plan_.type == VX355_JOIN_LEFT_SEMI_FILTER ||
      plan_.type == VX355_JOIN_COUNTING_LEFT_SEMI_FILTER || plan_.type == VX355_JOIN_RIGHT_SEMI_FILTER ||
      (plan_.type == VX355_JOIN_RIGHT_SEMI_PROJECT && !plan_.nullAware) || plan_.type == VX355_JOIN_RIGHT ||
      plan_.type == VX355_JOIN_RIGHT_ANTI;
  if (filterable && stats.num_distinct > 0 && stats.hash_mode != 0 /* kHash */ &&
      operatorCtx_->driverCtx()->queryConfig().hashProbeDynamicFilterPushdownEnabled()) {
    pushdownDynamicFilters();
  }
  return exec::BlockingReason::kNotBlocked;
}

void Vx355HashProbe::pushdownDynamicFilters() {
  auto* driver = operatorCtx_->driverCtx()->driver;
  const std::vector<column_index_t> keyChannels(plan_.probeKeys.begin(), plan_.probeKeys.end());
  driver->pushdownFilters(this, keyChannels, [&](column_index_t key, common::FilterPtr& filter) {
    vx355_key_filter described{};
    check(vx355_join_table_key_filter(table_, static_cast<int32_t>(key), &described));
    if (described.kind == VX355_KEY_FILTER_NONE) {
      return false;  // a key kind without value statistics (VectorHasher::getFilter returns nullptr)
    }
    if (!Vx355JoinTables::instance().firstToFilter(key_, static_cast<int32_t>(key))) {
      return true;  // a peer made it: the driver only installs the merged filter on this pipeline's scan
    }
    if (described.kind == VX355_KEY_FILTER_VALUES) {
      // VectorHasher::getFilter (exec/VectorHasher.cpp:731-780) -> common::createBigintValues: a BigintRange
      // when the values are consecutive, else a bitmask or a hash table
      std::vector<int64_t> values(static_cast<size_t>(std::max<int64_t>(described.num_distinct, 1)));
      int64_t count = 0;
      check(vx355_join_table_key_filter_values(
          table_, static_cast<int32_t>(key), values.data(), static_cast<int64_t>(values.size()), VX355_MEM_HOST, &count));
      values.resize(static_cast<size_t>(count));
      filter = common::createBigintValues(values, /*nullAllowed=*/false);
    } else {
      // more than kMaxDistinct values: the split-block Bloom filter of exec/HashTable.cpp:1133-1188, whose
      // blocks the library computes bit for bit as inserting every build value on the CPU would
      constexpr int32_t lanes = sizeof(SplitBlockBloomFilter::Block) / sizeof(uint32_t);
      auto bloom = std::make_shared<common::BigintValuesUsingBloomFilter>(described.num_distinct, /*nullAllowed=*/false);
      check(vx355_join_table_key_filter_bloom(
          table_, static_cast<int32_t>(key), lanes, reinterpret_cast<uint32_t*>(bloom->mutableBlocks()),
          common::BigintValuesUsingBloomFilter::numBlocks(described.num_distinct), VX355_MEM_HOST));
      addRuntimeStat("bloomFilterSize", RuntimeCounter(bloom->blocksByteSize()));  // HashProbe::kBloomFilterSize
      filter = std::move(bloom);
    }
    return true;
  });
}

void Vx355HashProbe::recordStats() {
  if (statsRecorded_ || handle_ == nullptr) {
    return;
  }
  statsRecorded_ = true;
  vx355_gpu_stats gpu{};
  if (vx355_join_probe_get_gpu_stats(handle_, &gpu) == VX355_OK) {
    recordGpuStats(*this, gpu, 0);
  }
}

bool Vx355HashProbe::needsInput() const {
  return handle_ != nullptr && !noMoreInput_ && inputDrained_;
}

void Vx355HashProbe::addInput(RowVectorPtr input) {
  input_ = std::move(input);
  decoded_ = std::make_unique<DecodedBatch>(*input_);
  check(vx355_join_probe_add_input_async(handle_, decoded_->get(), nullptr));  // (input_ / decoded_ keep the buffers)
  inputDrained_ = false;
}

void Vx355HashProbe::onPageDone(void* arg, int /*status*/, int32_t /*numRows*/, int32_t /*finished*/) {
  auto* page = static_cast<Page*>(arg);  // on the library's worker thread: wake the Driver, nothing else
  std::vector<ContinuePromise> promises;
  {
    std::lock_guard<std::mutex> lock(page->mutex);
    page->done.store(true, std::memory_order_release);
    promises.swap(page->promises);
  }
  for (auto& promise : promises) {
    promise.setValue();
  }
}

void Vx355HashProbe::startPage(bool buildSide) {
  const auto maxRows = outputBatchRows();
  auto page = std::make_unique<Page>();
  page->buildSide = buildSide;
  page->mapping = allocateIndices(maxRows, pool());
  page->buildRows.resize(maxRows);
  for (size_t j = 0; j < plan_.dependentOutputs.size(); ++j) {
    auto column = BaseVector::create(outputType_->childAt(plan_.dependentOutputs[j]), maxRows, pool());
    vx355_out_column c{};
    c.type_kind = plan_.dependentTypes[j];
    c.mem = VX355_MEM_HOST;
    c.values = column->values()->asMutable<void>();
    c.nulls = column->mutableRawNulls();
    page->out.push_back(c);
    page->ids.push_back(static_cast<int32_t>(j));
    page->buildColumns.push_back(std::move(column));
  }
  check(vx355_join_probe_get_output_async(
      handle_, buildSide ? 1 : 0, maxRows, page->mapping->asMutable<int32_t>(), page->buildRows.data(), VX355_MEM_HOST,
      page->out.data(), page->ids.data(), static_cast<int32_t>(page->ids.size()), &onPageDone, page.get(),
      &page->ticket));
  page_ = std::move(page);
}

void Vx355HashProbe::noMoreInput() {
  Operator::noMoreInput();
  if (!emitsBuildSide()) {
    return;
  }
  // the last prober lists the build rows nobody matched (exec/HashProbe.cpp:1189-1219)
  std::vector<ContinuePromise> promises;
  std::vector<std::shared_ptr<exec::Driver>> peers;
  lastProber_ = operatorCtx_->task()->allPeersFinished(planNodeId(), operatorCtx_->driver(), nullptr, promises, peers);
}

RowVectorPtr Vx355HashProbe::fillOutput(
    int32_t numRows,
    const BufferPtr& mapping,
    const int32_t* buildRows,
    std::vector<VectorPtr>& buildColumns,
    bool buildSide) {
  // HashProbe::fillOutput (exec/HashProbe.cpp:968-991): probe columns are the input's children
  // behind the mapping (no copy), build columns are what the library gathered
  std::vector<VectorPtr> children(outputType_->size());
  for (const auto& [probeChannel, outputChannel] : plan_.probeOutputs) {
    children[outputChannel] = buildSide
        ? BaseVector::createNullConstant(outputType_->childAt(outputChannel), numRows, pool())  // :1044-1048
        : exec::wrapChild(numRows, mapping, input_->childAt(probeChannel));
  }
  for (size_t j = 0; j < plan_.dependentOutputs.size(); ++j) {
    ownStrings(buildColumns[j], numRows);
    buildColumns[j]->resize(numRows);
    children[plan_.dependentOutputs[j]] = buildColumns[j];
  }
  if (plan_.matchOutput >= 0) {
    // LEFT_SEMI_PROJECT: build row >= 0 = TRUE, -1 = FALSE, -2 = NULL (null-aware IN)
    auto match = BaseVector::create<FlatVector<bool>>(BOOLEAN(), numRows, pool());
    for (int32_t i = 0; i < numRows; ++i) {
      if (buildRows[i] == -2) {
        match->setNull(i, true);
      } else {
        match->set(i, buildRows[i] >= 0);
      }
    }
    children[plan_.matchOutput] = match;
  }
  return std::make_shared<RowVector>(pool(), outputType_, nullptr, numRows, std::move(children));
}

RowVectorPtr Vx355HashProbe::getOutput() {
  if (handle_ == nullptr || finished_) {
    return nullptr;
  }
  const bool buildSide = wantsBuildSide();
  if (inputDrained_ && !buildSide) {
    if (noMoreInput_) {
      finished_ = true;
    }
    return nullptr;
  }
  if (page_ == nullptr) {
    startPage(buildSide);  // (a Driver that did not ask isBlocked() first)
  }
  if (!page_->done.load(std::memory_order_acquire)) {
    check(vx355_join_probe_wait(handle_));
    while (!page_->done.load(std::memory_order_acquire)) {
      std::this_thread::yield();  // (the callback runs right behind the ticket's completion)
    }
  }
  { std::lock_guard<std::mutex> callbackLeft(page_->mutex); }  // (the worker sets 'done' under this lock)
  auto page = std::move(page_);
  int32_t numRows = 0, done = 0;
  check(vx355_join_probe_output_result(handle_, page->ticket, &numRows, &done));
  if (page->buildSide) {
    buildSideDone_ = done != 0;
  } else {
    inputDrained_ = done != 0;
  }
  auto result = numRows > 0
      ? fillOutput(numRows, page->mapping, page->buildRows.data(), page->buildColumns, page->buildSide)
      : nullptr;
  if (inputDrained_) {
    input_.reset();
    decoded_.reset();
  }
  return result;
}

bool Vx355HashProbe::isFinished() {
  return finished_ || (noMoreInput_ && inputDrained_ && (!lastProber_ || buildSideDone_));
}

void Vx355HashProbe::close() {
  recordStats();
  if (handle_ != nullptr) {
    vx355_join_probe_destroy(handle_);
    handle_ = nullptr;
  }
  if (std::get<2>(key_) != "") {
    Vx355JoinTables::instance().removeProbe(key_);
    std::get<2>(key_) = "";
  }
  input_.reset();
  decoded_.reset();
  page_.reset();  // (the handle's worker is gone: no callback can come)
  Operator::close();
}

// ---- the adapter -------------------------------------------------------------------------------

namespace {

vx355_join_build_spec buildSpecOf(const JoinPlan& plan) {
  vx355_join_build_spec spec{};
  spec.num_keys = static_cast<int32_t>(plan.buildKeys.size());
  spec.key_cols = plan.buildKeys.data();
  spec.key_types = plan.buildKeyTypes.data();
  spec.num_dependents = static_cast<int32_t>(plan.dependentChannels.size());
  spec.dependent_cols = plan.dependentChannels.data();
  spec.dependent_types = plan.dependentTypes.data();
  spec.join_type = plan.type;
  spec.null_aware = plan.nullAware ? 1 : 0;
  spec.null_as_value = plan.nullAsValue ? 1 : 0;
  spec.drop_duplicates = plan.dropDuplicates ? 1 : 0;
  return spec;
}

}  // namespace

bool adaptJoins(const exec::DriverFactory& factory, exec::Driver& driver) {
  bool replaced = false;
  auto operators = driver.operators();
  for (int32_t i = 0; i < static_cast<int32_t>(operators.size()); ++i) {
    auto* build = dynamic_cast<exec::HashBuild*>(operators[i]);
    auto* probe = dynamic_cast<exec::HashProbe*>(operators[i]);
    if (build == nullptr && probe == nullptr) {
      continue;
    }
    auto node = joinNodeOf(factory, operators[i]->planNodeId());
    JoinPlan plan;
    if (node == nullptr || !toJoinPlan(*node, &plan)) {
      continue;
    }
    // Both pipelines of a join must decide alike, and the library has the last word
    // (VX355_EUNSUPPORTED at create): the first Driver of either pipeline asks it once with a trial
    // handle, the answer stays with the rendezvous entry of (task, split group, join node).
    const auto key = keyOf(driver.driverCtx(), node->id());
    const bool take = Vx355JoinTables::instance().accepted(key, [&] {
      const auto spec = buildSpecOf(plan);
      vx355_join_build* trial = nullptr;
      if (vx355_join_build_create(&spec, &trial) != VX355_OK) {
        return false;
      }
      vx355_join_build_destroy(trial);
      return true;
    });
    if (!take) {
      continue;
    }
    // FilterProject -> HashProbe fusion: the operator in front is the FilterProject of a FilterNode alone
    // (no projections: every output column is an input column) that feeds this probe, its conjunction is
    // in the library's class and the join kind lets a filtered-out row simply find nothing.
    int32_t begin = i;
    if (probe != nullptr && i > 0 && fusesInputFilter(plan)) {
      auto* filterProject = dynamic_cast<exec::FilterProject*>(operators[i - 1]);
      if (filterProject != nullptr && node->sources()[0]->id() == filterProject->planNodeId()) {
        std::shared_ptr<const core::FilterNode> filter;
        for (const auto& planNode : factory.planNodes) {
          if (planNode->id() == filterProject->planNodeId()) {
            filter = std::dynamic_pointer_cast<const core::FilterNode>(planNode);
          }
        }
        std::vector<vx355_filter_term> terms;
        if (filter != nullptr && toFilterTerms(*filter, &terms)) {
          plan.inputFilter = std::move(terms);
          begin = i - 1;
        }
      }
    }
    std::vector<std::unique_ptr<exec::Operator>> replacement;
    if (build != nullptr) {
      const auto spec = buildSpecOf(plan);
      vx355_join_build* handle = nullptr;
      check(vx355_join_build_create(&spec, &handle));  // the trial succeeded: a failure now is an error
      replacement.push_back(
          std::make_unique<Vx355HashBuild>(build->operatorId(), driver.driverCtx(), node, plan, handle));
    } else {
      replacement.push_back(
          std::make_unique<Vx355HashProbe>(operators[begin]->operatorId(), driver.driverCtx(), node, std::move(plan)));
    }
    factory.replaceOperators(driver, begin, i + 1, std::move(replacement));
    replaced = true;
    operators = driver.operators();
    i = begin;
  }
  return replaced;
}

}  // namespace facebook::velox::vx355
