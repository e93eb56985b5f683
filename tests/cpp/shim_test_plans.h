// TEST INFRASTRUCTURE: the reference's TPC-H Q1 plan (exec/tests/utils/TpchQueryBuilder.cpp:203-252)
// as core::PlanNodes of tests/velox_api_stub, the way PlanBuilder assembles them:
// function names as the DuckDB-based parser emits them (multiply / minus / plus / lte), DOUBLE
// literals, count(0) with a BIGINT constant, avg's intermediate type ROW(DOUBLE, BIGINT).
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "velox/core/Expressions.h"
#include "velox/core/PlanNode.h"

namespace shimtest {

using namespace facebook::velox;

constexpr int64_t kQ1Cutoff = 10471;  // DATE '1998-09-02', days since epoch

inline core::FieldAccessTypedExprPtr field(const RowTypePtr& type, const std::string& name) {
  return std::make_shared<core::FieldAccessTypedExpr>(type->findChild(name), name);
}
inline core::TypedExprPtr call(const std::string& name, TypePtr type, std::vector<core::TypedExprPtr> inputs) {
  return std::make_shared<core::CallTypedExpr>(std::move(type), std::move(inputs), name);
}
inline core::TypedExprPtr constant(TypePtr type, Variant value) {
  return std::make_shared<core::ConstantTypedExpr>(std::move(type), std::move(value));
}

inline std::shared_ptr<const core::ValuesNode> q1Scan() {
  return std::make_shared<core::ValuesNode>(
      "scan", ROW({"l_returnflag", "l_linestatus", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_shipdate"},
                  {VARCHAR(), VARCHAR(), DOUBLE(), DOUBLE(), DOUBLE(), DOUBLE(), DATE()}));
}

struct AggDef {
  std::string function;  // sum / avg / count / min / max ...
  std::string input;     // input column name ("" = no argument)
  TypePtr resultType;    // type of the call in THIS step
  TypePtr rawInputType;
};

inline std::shared_ptr<const core::AggregationNode> aggregation(
    const std::string& id, core::AggregationNode::Step step, const std::vector<std::string>& keys, const std::vector<AggDef>& defs,
    core::PlanNodePtr source, const std::vector<std::string>& names = {}) {
  const auto& inputType = source->outputType();
  std::vector<core::FieldAccessTypedExprPtr> keyExprs;
  for (const auto& key : keys) {
    keyExprs.push_back(field(inputType, key));
  }
  std::vector<core::AggregationNode::Aggregate> aggregates;
  std::vector<std::string> aggregateNames;
  for (size_t i = 0; i < defs.size(); ++i) {
    core::AggregationNode::Aggregate aggregate;
    std::vector<core::TypedExprPtr> inputs;
    if (!defs[i].input.empty()) {
      inputs.push_back(field(inputType, defs[i].input));
    }
    aggregate.call = std::make_shared<core::CallTypedExpr>(defs[i].resultType, inputs, defs[i].function);
    if (defs[i].rawInputType != nullptr) {
      aggregate.rawInputTypes = {defs[i].rawInputType};
    }
    aggregates.push_back(aggregate);
    aggregateNames.push_back(names.empty() ? "a" + std::to_string(i) : names[i]);
  }
  return std::make_shared<core::AggregationNode>(id, step, keyExprs, std::vector<core::FieldAccessTypedExprPtr>{}, aggregateNames,
                                                 aggregates, false, false, std::move(source));
}

inline std::shared_ptr<const core::AggregationNode> aggregationOverConstant(
    const std::string& id, const std::string& function, TypePtr type, Variant value, core::PlanNodePtr source,
    core::AggregationNode::Step step = core::AggregationNode::Step::kSingle, const std::vector<std::string>& keys = {}) {
  const auto& inputType = source->outputType();
  std::vector<core::FieldAccessTypedExprPtr> keyExprs;
  for (const auto& key : keys) {
    keyExprs.push_back(field(inputType, key));
  }
  core::AggregationNode::Aggregate aggregate;
  aggregate.call = std::make_shared<core::CallTypedExpr>(BIGINT(), std::vector<core::TypedExprPtr>{constant(type, std::move(value))}, function);
  aggregate.rawInputTypes = {type};
  return std::make_shared<core::AggregationNode>(id, step, keyExprs, std::vector<core::FieldAccessTypedExprPtr>{},
                                                 std::vector<std::string>{"a0"}, std::vector<core::AggregationNode::Aggregate>{aggregate},
                                                 false, false, std::move(source));
}

struct Q1Plan {
  std::shared_ptr<const core::ValuesNode> scan;
  std::shared_ptr<const core::FilterNode> filter;  // null when the filter is pushed into the scan
  std::shared_ptr<const core::ProjectNode> project;
  std::shared_ptr<const core::AggregationNode> partial;
  std::shared_ptr<const core::AggregationNode> final;
};

inline Q1Plan q1Plan(bool filterAsNode) {
  Q1Plan plan;
  plan.scan = q1Scan();
  const auto& scanType = plan.scan->outputType();
  core::PlanNodePtr source = plan.scan;
  if (filterAsNode) {
    // l_shipdate <= DATE '1998-09-02' (PlanBuilder::filtersAsNode)
    plan.filter = std::make_shared<core::FilterNode>(
        "filter", call("lte", BOOLEAN(), {field(scanType, "l_shipdate"), constant(DATE(), Variant(static_cast<int32_t>(kQ1Cutoff)))}),
        source);
    source = plan.filter;
  }
  auto price = field(scanType, "l_extendedprice");
  auto oneMinusDiscount = call("minus", DOUBLE(), {constant(DOUBLE(), Variant(1.0)), field(scanType, "l_discount")});
  auto onePlusTax = call("plus", DOUBLE(), {constant(DOUBLE(), Variant(1.0)), field(scanType, "l_tax")});
  auto discPrice = call("multiply", DOUBLE(), {price, oneMinusDiscount});
  auto charge = call("multiply", DOUBLE(), {discPrice, onePlusTax});
  plan.project = std::make_shared<core::ProjectNode>(
      "project",
      std::vector<std::string>{"l_returnflag", "l_linestatus", "l_quantity", "l_extendedprice", "l_sum_disc_price", "l_sum_charge",
                               "l_discount"},
      std::vector<core::TypedExprPtr>{field(scanType, "l_returnflag"), field(scanType, "l_linestatus"), field(scanType, "l_quantity"),
                                      price, discPrice, charge, field(scanType, "l_discount")},
      source);
  const auto avgIntermediate = ROW({"", ""}, {DOUBLE(), BIGINT()});
  // partial step: the calls carry the INTERMEDIATE types
  {
    const auto& inputType = plan.project->outputType();
    std::vector<core::AggregationNode::Aggregate> aggregates;
    auto add = [&](const std::string& fn, const std::string& column, TypePtr type) {
      core::AggregationNode::Aggregate a;
      a.call = std::make_shared<core::CallTypedExpr>(std::move(type), std::vector<core::TypedExprPtr>{field(inputType, column)}, fn);
      a.rawInputTypes = {DOUBLE()};
      aggregates.push_back(a);
    };
    add("sum", "l_quantity", DOUBLE());
    add("sum", "l_extendedprice", DOUBLE());
    add("sum", "l_sum_disc_price", DOUBLE());
    add("sum", "l_sum_charge", DOUBLE());
    add("avg", "l_quantity", avgIntermediate);
    add("avg", "l_extendedprice", avgIntermediate);
    add("avg", "l_discount", avgIntermediate);
    core::AggregationNode::Aggregate count;
    count.call = std::make_shared<core::CallTypedExpr>(
        BIGINT(), std::vector<core::TypedExprPtr>{constant(BIGINT(), Variant(static_cast<int64_t>(0)))}, "count");
    count.rawInputTypes = {BIGINT()};
    aggregates.push_back(count);
    std::vector<std::string> names;
    for (int i = 0; i < 8; ++i) {
      names.push_back("a" + std::to_string(i));
    }
    plan.partial = std::make_shared<core::AggregationNode>(
        "partial", core::AggregationNode::Step::kPartial,
        std::vector<core::FieldAccessTypedExprPtr>{field(inputType, "l_returnflag"), field(inputType, "l_linestatus")},
        std::vector<core::FieldAccessTypedExprPtr>{}, names, aggregates, false, false, plan.project);
  }
  // final step (PlanBuilder::finalAggregation): same functions over the partial step's output
  // columns, raw input types kept, result types final
  {
    const auto& inputType = plan.partial->outputType();
    std::vector<core::AggregationNode::Aggregate> aggregates;
    std::vector<std::string> names;
    const char* fns[8] = {"sum", "sum", "sum", "sum", "avg", "avg", "avg", "count"};
    for (int i = 0; i < 8; ++i) {
      core::AggregationNode::Aggregate a;
      const std::string column = "a" + std::to_string(i);
      a.call = std::make_shared<core::CallTypedExpr>(i == 7 ? BIGINT() : DOUBLE(),
                                                     std::vector<core::TypedExprPtr>{field(inputType, column)}, fns[i]);
      a.rawInputTypes = {i == 7 ? BIGINT() : DOUBLE()};
      aggregates.push_back(a);
      names.push_back(column);
    }
    plan.final = std::make_shared<core::AggregationNode>(
        "final", core::AggregationNode::Step::kFinal,
        std::vector<core::FieldAccessTypedExprPtr>{field(inputType, "l_returnflag"), field(inputType, "l_linestatus")},
        std::vector<core::FieldAccessTypedExprPtr>{}, names, aggregates, false, false, plan.partial);
  }
  return plan;
}

}  // namespace shimtest
