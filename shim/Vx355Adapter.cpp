// See Vx355Adapter.h. Compiled on the Velox side only.
#include "Vx355Adapter.h"
#include "Vx355JoinAdapter.h"

#include <algorithm>
#include <string>

#include "velox/core/PlanNode.h"
#include "velox/core/QueryConfig.h"
#include "velox/exec/Aggregate.h"
#include "velox/exec/HashAggregation.h"
#include "velox/exec/Task.h"
#include "velox/vector/FlatVector.h"

namespace facebook::velox::vx355 {

namespace {

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

// vx355_agg_spec of an AggregationNode: keys, aggregates, step, ignoreNullKeys
// (core/PlanNode.h:1120-1370). Returns false when the plan is outside what the library takes -
// the CPU operator then stays in place, like cuDF's adapter does (ToCudf.cpp:230-242).
struct AggSpec {
  std::vector<int32_t> keyCols, keyTypes;
  std::vector<vx355_agg_fn> fns;
  vx355_agg_spec c{};
};

bool toAggSpec(const core::AggregationNode& node, AggSpec* out) {
  const auto& inputType = node.sources()[0]->outputType();
  for (const auto& key : node.groupingKeys()) {
    const auto channel = exec::exprToChannel(key.get(), inputType);
    if (channel == kConstantChannel || !scalarKind(inputType->childAt(channel)->kind())) {
      return false;
    }
    out->keyCols.push_back(static_cast<int32_t>(channel));
    out->keyTypes.push_back(static_cast<int32_t>(inputType->childAt(channel)->kind()));
  }
  const bool raw = node.step() == core::AggregationNode::Step::kPartial ||
      node.step() == core::AggregationNode::Step::kSingle;
  for (const auto& aggregate : node.aggregates()) {
    if (!aggregate.sortingKeys.empty()) {
      // (sum / count / min / max / avg do not depend on the order: the reference drops it too,
      // exec/AggregateInfo.cpp:124-138; anything else is not ours)
    }
    vx355_agg_fn fn{};
    const auto& name = aggregate.call->name();
    if (name == "sum") {
      fn.kind = VX355_AGG_SUM;
    } else if (name == "count") {
      fn.kind = aggregate.call->inputs().empty() ? VX355_AGG_COUNT_STAR : VX355_AGG_COUNT;
    } else if (name == "min") {
      fn.kind = VX355_AGG_MIN;
    } else if (name == "max") {
      fn.kind = VX355_AGG_MAX;
    } else if (name == "avg") {
      fn.kind = VX355_AGG_AVG;
    } else {
      return false;
    }
    fn.input_col = fn.input_col2 = fn.mask_col = -1;
    fn.input_type = VX355_BIGINT;
    if (!aggregate.call->inputs().empty()) {
      const auto channel = exec::exprToChannel(aggregate.call->inputs()[0].get(), inputType);
      if (channel == kConstantChannel) {
        return false;
      }
      fn.input_col = static_cast<int32_t>(channel);
      const auto& type = raw ? inputType->childAt(channel) : aggregate.rawInputTypes[0];
      if (!scalarKind(type->kind())) {
        return false;
      }
      fn.input_type = static_cast<int32_t>(type->kind());
      if (fn.kind == VX355_AGG_AVG && !raw) {
        // the intermediate ROW(DOUBLE sum, BIGINT count) arrives flattened by the shim's
        // DecodedBatch: the count child is the next vx355 column (see DecodedBatch)
        fn.input_col2 = fn.input_col + 1;
      }
    }
    if (aggregate.mask) {
      fn.mask_col = static_cast<int32_t>(exec::exprToChannel(aggregate.mask.get(), inputType));
    }
    fn.flags = aggregate.distinct ? VX355_AGG_FN_DISTINCT : 0;
    out->fns.push_back(fn);
  }
  out->c.num_keys = static_cast<int32_t>(out->keyCols.size());
  out->c.key_cols = out->keyCols.data();
  out->c.key_types = out->keyTypes.data();
  out->c.num_aggs = static_cast<int32_t>(out->fns.size());
  out->c.aggs = out->fns.data();
  out->c.step = static_cast<int32_t>(node.step());  // same numeric values (core/PlanNode.h:1122-1131)
  out->c.ignore_null_keys = node.ignoreNullKeys() ? 1 : 0;
  return true;
}

std::shared_ptr<const core::PlanNode> planNodeOf(const exec::DriverFactory& factory, const core::PlanNodeId& id) {
  for (const auto& node : factory.planNodes) {
    if (node->id() == id) {
      return node;
    }
  }
  return factory.consumerNode && factory.consumerNode->id() == id ? factory.consumerNode : nullptr;
}

bool adapt(const exec::DriverFactory& factory, exec::Driver& driver) {
  bool replaced = adaptJoins(factory, driver);  // HashBuild / HashProbe: Vx355JoinAdapter.cpp
  auto operators = driver.operators();
  for (int32_t i = 0; i < static_cast<int32_t>(operators.size()); ++i) {
    auto* aggregation = dynamic_cast<exec::HashAggregation*>(operators[i]);
    if (aggregation == nullptr) {
      continue;
    }
    auto node = std::dynamic_pointer_cast<const core::AggregationNode>(planNodeOf(factory, aggregation->planNodeId()));
    AggSpec spec;
    if (node == nullptr || !node->preGroupedKeys().empty() || !toAggSpec(*node, &spec)) {
      continue;
    }
    vx355_agg* handle = nullptr;
    if (vx355_agg_create(&spec.c, &handle) != VX355_OK) {
      continue;  // VX355_EUNSUPPORTED: the CPU operator stays
    }
    std::vector<std::unique_ptr<exec::Operator>> replacement;
    replacement.push_back(
        std::make_unique<Vx355HashAggregation>(aggregation->operatorId(), driver.driverCtx(), node, handle));
    factory.replaceOperators(driver, i, i + 1, std::move(replacement));
    replaced = true;
  }
  return replaced;
}

}  // namespace

void registerVx355(int device) {
  VELOX_CHECK_EQ(vx355_init(device), VX355_OK, "{}", vx355_last_error());
  exec::DriverFactory::registerAdapter(exec::DriverAdapter{"vx355", /*inspect=*/{}, adapt});
}

// ---- batches in, columns out -------------------------------------------------------------------

DecodedBatch::DecodedBatch(const RowVector& input) {
  const auto numRows = input.size();
  decoded_.reserve(input.childrenSize());
  columns_.reserve(input.childrenSize());
  for (size_t i = 0; i < input.childrenSize(); ++i) {
    const auto& child = input.childAt(i);
    decoded_.emplace_back(*child->loadedVector());  // all rows
    const auto& d = decoded_.back();
    vx355_column col{};
    col.type_kind = static_cast<int32_t>(child->typeKind());
    col.mem = VX355_MEM_HOST;
    col.values = d.data<void>();
    col.nulls = d.nulls();  // bit per top-level row after decoding, 1 = valid: vx355's polarity
    if (d.isConstantMapping()) {
      col.encoding = VX355_CONSTANT;
      // the single value sits at index(0) of the base
      col.values = static_cast<const char*>(d.data<void>()) + static_cast<size_t>(d.index(0)) * child->type()->cppSizeInBytes();
      col.nulls = nullptr;
      if (d.isNullAt(0)) {
        static const uint64_t kNull = 0;
        col.nulls = &kNull;
      }
    } else if (d.isIdentityMapping()) {
      col.encoding = VX355_FLAT;
    } else {
      col.encoding = VX355_DICTIONARY;
      col.indices = d.indices();
      col.base_size = d.base()->size();
    }
    columns_.push_back(col);
  }
  batch_.num_rows = numRows;
  batch_.num_cols = static_cast<int32_t>(columns_.size());
  batch_.cols = columns_.data();
}

OutColumns::OutColumns(RowVector& result) {
  for (size_t i = 0; i < result.childrenSize(); ++i) {
    auto& child = result.childAt(i);
    vx355_out_column col{};
    col.type_kind = static_cast<int32_t>(child->typeKind());
    col.mem = VX355_MEM_HOST;
    col.values = child->values() ? child->values()->asMutable<void>() : nullptr;  // flat scalar children of the result
    col.nulls = child->mutableRawNulls();
    columns_.push_back(col);
  }
}

void ownStrings(const VectorPtr& column, vector_size_t numRows) {
  if (column->typeKind() != TypeKind::VARCHAR && column->typeKind() != TypeKind::VARBINARY) {
    return;
  }
  auto* flat = column->asFlatVector<StringView>();
  for (vector_size_t i = 0; i < numRows; ++i) {
    if (!flat->isNullAt(i) && !flat->valueAt(i).isInline()) {
      flat->set(i, flat->valueAt(i));  // copies the bytes into a string buffer of the vector
    }
  }
}

// ---- the operator ------------------------------------------------------------------------------

Vx355HashAggregation::Vx355HashAggregation(
    int32_t operatorId,
    exec::DriverCtx* driverCtx,
    const std::shared_ptr<const core::AggregationNode>& node,
    vx355_agg* handle)
    : Operator(driverCtx, node->outputType(), operatorId, node->id(), "Vx355HashAggregation"),
      handle_(handle),
      isPartialOutput_(exec::isPartialOutput(node->step())),
      isGlobal_(node->groupingKeys().empty()),
      maxPartialMemory_(driverCtx->queryConfig().maxPartialAggregationMemoryUsage()) {}

Vx355HashAggregation::~Vx355HashAggregation() {
  if (handle_ != nullptr) {
    vx355_agg_destroy(handle_);
  }
}

void Vx355HashAggregation::check(int status) {
  if (status == VX355_OK) {
    return;
  }
  if (status == VX355_EUSER) {
    VELOX_USER_FAIL("{}", vx355_last_error());  // e.g. "integer overflow" (sum(BIGINT))
  }
  VELOX_FAIL("{}", vx355_last_error());
}

void Vx355HashAggregation::releaseCompleted() {
  int64_t submitted = 0, completed = 0;
  check(vx355_agg_poll(handle_, &submitted, &completed));
  inFlight_.erase(
      std::remove_if(inFlight_.begin(), inFlight_.end(), [&](const InFlight& b) { return b.ticket <= completed; }),
      inFlight_.end());
}

bool Vx355HashAggregation::needsInput() const {
  return !noMoreInput_ && !flushing_;
}

exec::BlockingReason Vx355HashAggregation::isBlocked(ContinueFuture* /*future*/) {
  // The library takes whole chunks of ~1 M rows (parallel ingest): a few thousand vectors may be in
  // flight. Bound the memory they pin; a Driver that finds the operator "blocked" simply comes back
  // (kYield-style polling keeps the shim free of callbacks into the library's worker thread).
  releaseCompleted();
  constexpr size_t kMaxInFlight = 4096;
  if (inFlight_.size() >= kMaxInFlight) {
    check(vx355_agg_wait(handle_));
    releaseCompleted();
  }
  return exec::BlockingReason::kNotBlocked;
}

void Vx355HashAggregation::addInput(RowVectorPtr input) {
  auto decoded = std::make_unique<DecodedBatch>(*input);
  int64_t ticket = 0;
  check(vx355_agg_add_input_async(handle_, decoded->get(), &ticket));
  inFlight_.push_back(InFlight{ticket, std::move(input), std::move(decoded)});
  if (isPartialOutput_ && !isGlobal_ && (ticket & 63) == 0 && partialFull()) {
    // HashAggregation.cpp:191-236,293-327: flush the partial groups, continue with an empty table
    check(vx355_agg_flush(handle_));
    flushing_ = true;
  }
}

bool Vx355HashAggregation::partialFull() {
  vx355_agg_stats stats{};
  check(vx355_agg_get_stats(handle_, &stats));
  return stats.table_bytes > maxPartialMemory_;
}

void Vx355HashAggregation::noMoreInput() {
  Operator::noMoreInput();
  check(vx355_agg_no_more_input(handle_));  // waits for the queued batches
  inFlight_.clear();
}

RowVectorPtr Vx355HashAggregation::getOutput() {
  if (finished_ || (!noMoreInput_ && !flushing_)) {
    return nullptr;
  }
  const auto maxRows = outputBatchRows();
  auto result = std::static_pointer_cast<RowVector>(BaseVector::create(outputType_, maxRows, pool()));
  for (auto& child : result->children()) {
    child->mutableRawNulls();  // the library writes validity for every column
  }
  OutColumns out(*result);
  int32_t numRows = 0, finished = 0;
  check(vx355_agg_get_output(handle_, out.data(), out.size(), maxRows, &numRows, &finished));
  if (finished) {
    if (flushing_) {
      flushing_ = false;  // the table is empty again: GroupingSet::resetTable
    } else {
      finished_ = true;
    }
  }
  if (numRows == 0) {
    return nullptr;
  }
  for (auto& child : result->children()) {
    ownStrings(child, numRows);
  }
  result->resize(numRows);
  return result;
}

bool Vx355HashAggregation::isFinished() {
  return finished_;
}

void Vx355HashAggregation::close() {
  if (handle_ != nullptr) {
    vx355_agg_destroy(handle_);
    handle_ = nullptr;
  }
  inFlight_.clear();
  Operator::close();
}

}  // namespace facebook::velox::vx355
