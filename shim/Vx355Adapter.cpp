// See Vx355Adapter.h. Compiled on the Velox side (and, in this repository, against
// tests/velox_api_stub by tests/test_shim.py).
#include "Vx355Adapter.h"
#include "Vx355JoinAdapter.h"

#include <algorithm>
#include <thread>
#include <cstring>
#include <string>

#include "velox/core/Expressions.h"
#include "velox/core/QueryConfig.h"
#include "velox/exec/Aggregate.h"
#include "velox/exec/FilterProject.h"
#include "velox/exec/HashAggregation.h"
#include "velox/exec/Task.h"
#include "velox/common/memory/MemoryArbitrator.h"
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

bool integerLike(TypeKind kind) {
  return kind == TypeKind::TINYINT || kind == TypeKind::SMALLINT || kind == TypeKind::INTEGER || kind == TypeKind::BIGINT;
}

bool structOfScalars(const TypePtr& type) {
  if (!type->isRow() || type->size() == 0) {
    return false;
  }
  for (uint32_t i = 0; i < type->size(); ++i) {
    if (!scalarKind(type->childAt(i)->kind())) {
      return false;
    }
  }
  return true;
}

// "presto.default.plus" -> "plus" (function names may carry a registration prefix)
std::string baseName(const std::string& name) {
  const auto dot = name.rfind('.');
  return dot == std::string::npos ? name : name.substr(dot + 1);
}

// A constant of an expression as ConstantColumn / filter-term operand. false: a kind the library
// does not take as a constant.
bool constantValue(const core::ConstantTypedExpr& expr, ConstantColumn* out) {
  const auto kind = expr.type()->kind();
  if (!scalarKind(kind) || kind == TypeKind::TIMESTAMP) {
    return false;
  }
  out->typeKind = static_cast<int32_t>(kind);
  out->isNull = expr.isNull();
  if (out->isNull) {
    return true;
  }
  if (expr.hasValueVector()) {
    // a constant vector: read row 0 through DecodedVector
    DecodedVector decoded(*expr.valueVector());
    const auto index = decoded.index(0);
    if (kind == TypeKind::BOOLEAN) {
      out->value[0] = bits::isBitSet(decoded.data<uint64_t>(), index) ? 1 : 0;
      return true;
    }
    const auto width = expr.type()->cppSizeInBytes();
    std::memcpy(out->value, decoded.data<char>() + static_cast<size_t>(index) * width, width);
    if (kind == TypeKind::VARCHAR || kind == TypeKind::VARBINARY) {
      StringView view;
      std::memcpy(&view, out->value, sizeof(view));
      if (!view.isInline()) {
        out->longString.assign(view.data(), view.size());
      }
    }
    return true;
  }
  const auto& v = expr.value();
  switch (kind) {
    case TypeKind::BOOLEAN: {
      out->value[0] = v.value<TypeKind::BOOLEAN>() ? 1 : 0;
      return true;
    }
    case TypeKind::TINYINT: {
      const auto x = v.value<TypeKind::TINYINT>();
      std::memcpy(out->value, &x, sizeof(x));
      return true;
    }
    case TypeKind::SMALLINT: {
      const auto x = v.value<TypeKind::SMALLINT>();
      std::memcpy(out->value, &x, sizeof(x));
      return true;
    }
    case TypeKind::INTEGER: {
      const auto x = v.value<TypeKind::INTEGER>();
      std::memcpy(out->value, &x, sizeof(x));
      return true;
    }
    case TypeKind::BIGINT: {
      const auto x = v.value<TypeKind::BIGINT>();
      std::memcpy(out->value, &x, sizeof(x));
      return true;
    }
    case TypeKind::REAL: {
      const auto x = v.value<TypeKind::REAL>();
      std::memcpy(out->value, &x, sizeof(x));
      return true;
    }
    case TypeKind::DOUBLE: {
      const auto x = v.value<TypeKind::DOUBLE>();
      std::memcpy(out->value, &x, sizeof(x));
      return true;
    }
    case TypeKind::VARCHAR:
    case TypeKind::VARBINARY: {
      out->longString = kind == TypeKind::VARCHAR ? v.value<TypeKind::VARCHAR>() : v.value<TypeKind::VARBINARY>();
      return true;  // the view is made where the column is (DecodedBatch): it points into longString
    }
    default:
      return false;
  }
}

int64_t constantAsInt64(const ConstantColumn& c) {
  switch (static_cast<TypeKind>(c.typeKind)) {
    case TypeKind::BOOLEAN:
    case TypeKind::TINYINT: {
      int8_t x;
      std::memcpy(&x, c.value, sizeof(x));
      return x;
    }
    case TypeKind::SMALLINT: {
      int16_t x;
      std::memcpy(&x, c.value, sizeof(x));
      return x;
    }
    case TypeKind::INTEGER: {
      int32_t x;
      std::memcpy(&x, c.value, sizeof(x));
      return x;
    }
    default: {
      int64_t x;
      std::memcpy(&x, c.value, sizeof(x));
      return x;
    }
  }
}

double constantAsDouble(const ConstantColumn& c) {
  if (static_cast<TypeKind>(c.typeKind) == TypeKind::REAL) {
    float x;
    std::memcpy(&x, c.value, sizeof(x));
    return x;
  }
  double x;
  std::memcpy(&x, c.value, sizeof(x));
  return x;
}

const core::FieldAccessTypedExpr* asInputColumn(const core::TypedExprPtr& expr) {
  auto* field = dynamic_cast<const core::FieldAccessTypedExpr*>(expr.get());
  if (field == nullptr) {
    return nullptr;
  }
  // a plain column of the input row (exec/FilterProject.cpp:25-41 has the same test)
  if (!field->inputs().empty() &&
      !(field->inputs().size() == 1 && dynamic_cast<const core::InputTypedExpr*>(field->inputs()[0].get()))) {
    return nullptr;
  }
  return field;
}

// ---- filter: conjunction of column-vs-constant comparisons ---------------------------------------

int32_t flipped(int32_t cmp) {
  switch (cmp) {
    case VX355_CMP_LT:
      return VX355_CMP_GT;
    case VX355_CMP_LE:
      return VX355_CMP_GE;
    case VX355_CMP_GT:
      return VX355_CMP_LT;
    case VX355_CMP_GE:
      return VX355_CMP_LE;
    default:
      return cmp;
  }
}

bool comparisonOf(const std::string& name, int32_t* cmp) {
  static const std::pair<const char*, int32_t> kNames[] = {
      {"eq", VX355_CMP_EQ}, {"neq", VX355_CMP_NE}, {"lt", VX355_CMP_LT},
      {"lte", VX355_CMP_LE}, {"gt", VX355_CMP_GT}, {"gte", VX355_CMP_GE}};
  for (const auto& [n, c] : kNames) {
    if (name == n) {
      *cmp = c;
      return true;
    }
  }
  return false;
}

bool makeTerm(const core::FieldAccessTypedExpr& column, int32_t cmp, const core::ConstantTypedExpr& constant,
              const RowTypePtr& scanType, const ColumnLayout& layout, vx355_filter_term* term) {
  const auto channel = scanType->getChildIdxIfExists(column.name());
  if (!channel.has_value()) {
    return false;
  }
  const auto kind = scanType->childAt(*channel)->kind();
  ConstantColumn value;
  if (constant.type()->kind() != kind || !constantValue(constant, &value) || value.isNull) {
    return false;  // (a null constant makes the whole filter false: not worth a kernel - stays on the CPU)
  }
  *term = vx355_filter_term{};
  term->col = layout.first[*channel];
  term->cmp = cmp;
  if (integerLike(kind)) {
    term->const_kind = VX355_BIGINT;
    term->i64 = constantAsInt64(value);
  } else if (kind == TypeKind::DOUBLE || kind == TypeKind::REAL) {
    term->const_kind = VX355_DOUBLE;
    term->f64 = constantAsDouble(value);
  } else if (kind == TypeKind::VARCHAR || kind == TypeKind::VARBINARY) {
    std::string bytes = value.longString;
    if (bytes.empty()) {
      StringView view;
      std::memcpy(&view, value.value, sizeof(view));
      bytes.assign(view.data(), view.size());
    }
    if (bytes.size() > 12 || (cmp != VX355_CMP_EQ && cmp != VX355_CMP_NE)) {
      return false;
    }
    term->const_kind = VX355_VARCHAR;
    term->str_size = static_cast<int32_t>(bytes.size());
    std::memcpy(term->str, bytes.data(), bytes.size());
  } else {
    return false;
  }
  return true;
}

bool addFilterTerms(const core::TypedExprPtr& expr, const RowTypePtr& scanType, const ColumnLayout& layout,
                    std::vector<vx355_filter_term>* terms) {
  auto* call = dynamic_cast<const core::CallTypedExpr*>(expr.get());
  if (call == nullptr) {
    return false;
  }
  const auto name = baseName(call->name());
  if (name == "and") {
    for (const auto& input : call->inputs()) {
      if (!addFilterTerms(input, scanType, layout, terms)) {
        return false;
      }
    }
    return true;
  }
  vx355_filter_term term;
  if (name == "between" && call->inputs().size() == 3) {
    auto* column = asInputColumn(call->inputs()[0]);
    auto* lo = dynamic_cast<const core::ConstantTypedExpr*>(call->inputs()[1].get());
    auto* hi = dynamic_cast<const core::ConstantTypedExpr*>(call->inputs()[2].get());
    if (column == nullptr || lo == nullptr || hi == nullptr) {
      return false;
    }
    if (!makeTerm(*column, VX355_CMP_GE, *lo, scanType, layout, &term)) {
      return false;
    }
    terms->push_back(term);
    if (!makeTerm(*column, VX355_CMP_LE, *hi, scanType, layout, &term)) {
      return false;
    }
    terms->push_back(term);
    return true;
  }
  int32_t cmp = 0;
  if (!comparisonOf(name, &cmp) || call->inputs().size() != 2) {
    return false;
  }
  auto* leftColumn = asInputColumn(call->inputs()[0]);
  auto* rightColumn = asInputColumn(call->inputs()[1]);
  auto* leftConstant = dynamic_cast<const core::ConstantTypedExpr*>(call->inputs()[0].get());
  auto* rightConstant = dynamic_cast<const core::ConstantTypedExpr*>(call->inputs()[1].get());
  if (leftColumn != nullptr && rightConstant != nullptr) {
    if (!makeTerm(*leftColumn, cmp, *rightConstant, scanType, layout, &term)) {
      return false;
    }
  } else if (leftConstant != nullptr && rightColumn != nullptr) {
    if (!makeTerm(*rightColumn, flipped(cmp), *leftConstant, scanType, layout, &term)) {
      return false;
    }
  } else {
    return false;
  }
  terms->push_back(term);
  return true;
}

// ---- projections: f0 * f1 * ... with f = scale * column + offset, all DOUBLE ----------------------

bool doubleConstant(const core::TypedExprPtr& expr, double* out) {
  auto* constant = dynamic_cast<const core::ConstantTypedExpr*>(expr.get());
  ConstantColumn value;
  if (constant == nullptr || constant->type()->kind() != TypeKind::DOUBLE || !constantValue(*constant, &value) || value.isNull) {
    return false;
  }
  *out = constantAsDouble(value);
  return true;
}

bool doubleColumn(const core::TypedExprPtr& expr, const RowTypePtr& scanType, const ColumnLayout& layout, int32_t* col) {
  auto* column = asInputColumn(expr);
  if (column == nullptr) {
    return false;
  }
  const auto channel = scanType->getChildIdxIfExists(column->name());
  if (!channel.has_value() || scanType->childAt(*channel)->kind() != TypeKind::DOUBLE) {
    return false;  // Velox multiplies REAL in float and integers with overflow checks: not this class
  }
  *col = layout.first[*channel];
  return true;
}

// One factor. The library computes scale * x + offset without contraction:
//   x        = 1 * x + (-0.0)   (adding -0.0 keeps the sign of a zero)
//   c - x    = -1 * x + c       (IEEE subtraction is the addition of the negation)
//   x + c, c + x = 1 * x + c;  x - c = 1 * x + (-c)
bool makeFactor(const core::TypedExprPtr& expr, const RowTypePtr& scanType, const ColumnLayout& layout, vx355_factor* out) {
  *out = vx355_factor{};
  double constant = 0;
  int32_t col = -1;
  if (doubleColumn(expr, scanType, layout, &col)) {
    *out = vx355_factor{col, 0, 1.0, -0.0};
    return true;
  }
  if (doubleConstant(expr, &constant)) {
    *out = vx355_factor{-1, 0, 0.0, constant};
    return true;
  }
  auto* call = dynamic_cast<const core::CallTypedExpr*>(expr.get());
  if (call == nullptr || call->inputs().size() != 2 || call->type()->kind() != TypeKind::DOUBLE) {
    return false;
  }
  const auto name = baseName(call->name());
  const auto& a = call->inputs()[0];
  const auto& b = call->inputs()[1];
  if (name == "plus") {
    if (doubleColumn(a, scanType, layout, &col) && doubleConstant(b, &constant)) {
      *out = vx355_factor{col, 0, 1.0, constant};
      return true;
    }
    if (doubleConstant(a, &constant) && doubleColumn(b, scanType, layout, &col)) {
      *out = vx355_factor{col, 0, 1.0, constant};
      return true;
    }
  } else if (name == "minus" || name == "subtract") {
    if (doubleColumn(a, scanType, layout, &col) && doubleConstant(b, &constant)) {
      *out = vx355_factor{col, 0, 1.0, -constant};
      return true;
    }
    if (doubleConstant(a, &constant) && doubleColumn(b, scanType, layout, &col)) {
      *out = vx355_factor{col, 0, -1.0, constant};
      return true;
    }
  }
  return false;
}

// Left-deep products only: ((f0 * f1) * f2) * f3 is what the library evaluates, and a * (b * c)
// rounds differently.
bool makeProjection(const core::TypedExprPtr& expr, const RowTypePtr& scanType, const ColumnLayout& layout,
                    vx355_projection* out) {
  std::vector<core::TypedExprPtr> factors;
  core::TypedExprPtr cursor = expr;
  while (true) {
    auto* call = dynamic_cast<const core::CallTypedExpr*>(cursor.get());
    if (call != nullptr && baseName(call->name()) == "multiply" && call->inputs().size() == 2 &&
        call->type()->kind() == TypeKind::DOUBLE) {
      factors.push_back(call->inputs()[1]);
      cursor = call->inputs()[0];
      continue;
    }
    factors.push_back(cursor);
    break;
  }
  if (factors.size() > 4) {
    return false;
  }
  std::reverse(factors.begin(), factors.end());
  *out = vx355_projection{};
  out->num_factors = static_cast<int32_t>(factors.size());
  for (size_t i = 0; i < factors.size(); ++i) {
    if (!makeFactor(factors[i], scanType, layout, &out->factors[i])) {
      return false;
    }
  }
  return true;
}

std::vector<InputBinding> identityBindings(const RowTypePtr& type, const ColumnLayout& layout) {
  std::vector<InputBinding> bindings(type->size());
  for (uint32_t c = 0; c < type->size(); ++c) {
    bindings[c] = InputBinding{layout.first[c], type->childAt(c)};
  }
  return bindings;
}

template <typename T>
std::shared_ptr<const T> planNodeOf(const exec::DriverFactory& factory, const core::PlanNodeId& id) {
  for (const auto& node : factory.planNodes) {
    if (node->id() == id) {
      return std::dynamic_pointer_cast<const T>(node);
    }
  }
  return factory.consumerNode && factory.consumerNode->id() == id
      ? std::dynamic_pointer_cast<const T>(factory.consumerNode)
      : nullptr;
}

}  // namespace

// ---- plan translation --------------------------------------------------------------------------

ColumnLayout::ColumnLayout(const RowTypePtr& type) {
  for (uint32_t c = 0; c < type->size(); ++c) {
    first.push_back(numColumns);
    count.push_back(structOfScalars(type->childAt(c)) ? static_cast<int32_t>(type->childAt(c)->size()) : 1);
    numColumns += count.back();
  }
}

bool toFusedInput(const core::FilterNode* filter, const core::ProjectNode* project, FusedInput* out) {
  if (filter == nullptr && project == nullptr) {
    return false;
  }
  out->scanType = filter != nullptr ? filter->sources()[0]->outputType() : project->sources()[0]->outputType();
  const ColumnLayout layout(out->scanType);
  if (filter != nullptr && !addFilterTerms(filter->filter(), out->scanType, layout, &out->terms)) {
    return false;
  }
  if (out->terms.size() > 4) {
    return false;
  }
  if (project == nullptr) {
    out->bindings = identityBindings(out->scanType, layout);
    return true;
  }
  for (const auto& expr : project->projections()) {
    if (auto* column = asInputColumn(expr)) {
      const auto channel = out->scanType->getChildIdxIfExists(column->name());
      if (!channel.has_value()) {
        return false;
      }
      out->bindings.push_back(InputBinding{layout.first[*channel], out->scanType->childAt(*channel)});
      continue;
    }
    vx355_projection projection;
    if (out->projections.size() < 4 && makeProjection(expr, out->scanType, layout, &projection)) {
      out->bindings.push_back(
          InputBinding{VX355_PROJECTION_COL_BASE + static_cast<int32_t>(out->projections.size()), DOUBLE()});
      out->projections.push_back(projection);
      continue;
    }
    // a projection outside the class: fine as long as the aggregation does not read it
    out->bindings.push_back(InputBinding{-1, expr->type()});
  }
  return true;
}

bool toFilterTerms(const core::FilterNode& filter, std::vector<vx355_filter_term>* out) {
  const auto& scanType = filter.sources()[0]->outputType();
  const ColumnLayout layout(scanType);
  if (layout.numColumns != static_cast<int32_t>(scanType->size())) {
    return false;  // a struct channel would renumber the columns the join's plan refers to
  }
  return addFilterTerms(filter.filter(), scanType, layout, out) && out->size() <= 4;
}

bool toAggSpec(const core::AggregationNode& node, const std::vector<InputBinding>& bindings, const ColumnLayout& layout,
               AggSpec* out) {
  const auto& inputType = node.sources()[0]->outputType();
  if (bindings.size() != inputType->size() || !node.preGroupedKeys().empty() || !node.globalGroupingSets().empty()) {
    return false;
  }
  out->layout = layout;
  for (const auto& key : node.groupingKeys()) {
    const auto channel = exec::exprToChannel(key.get(), inputType);
    if (channel == kConstantChannel || !scalarKind(inputType->childAt(channel)->kind()) ||
        bindings[channel].column < 0 || bindings[channel].column >= VX355_PROJECTION_COL_BASE) {
      return false;  // keys are plain scalar columns of the batch
    }
    out->keyCols.push_back(bindings[channel].column);
    out->keyTypes.push_back(static_cast<int32_t>(inputType->childAt(channel)->kind()));
  }
  const bool raw = exec::isRawInput(node.step());
  for (const auto& aggregate : node.aggregates()) {
    // (ORDER BY inside sum / count / min / max / avg: the reference drops the sorting keys itself,
    // exec/AggregateInfo.cpp:124-138)
    vx355_agg_fn fn{};
    const auto name = baseName(aggregate.call->name());
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
    if (aggregate.call->inputs().size() > 1) {
      return false;
    }
    if (!aggregate.call->inputs().empty()) {
      const auto& argument = aggregate.call->inputs()[0];
      const auto channel = exec::exprToChannel(argument.get(), inputType);
      if (channel == kConstantChannel) {
        // exec/AggregateInfo.cpp:62-69: a constant argument reaches the function as a constant
        // vector. count(0) of TPC-H Q1 (TpchQueryBuilder.cpp:243) counts every row like count(*);
        // in general the constant travels as one more (VX355_CONSTANT) column of every batch.
        if (!raw) {
          return false;  // intermediate inputs are columns
        }
        ConstantColumn constant;
        if (!constantValue(*dynamic_cast<const core::ConstantTypedExpr*>(argument.get()), &constant)) {
          return false;
        }
        if (fn.kind == VX355_AGG_COUNT && !constant.isNull) {
          fn.kind = VX355_AGG_COUNT_STAR;
        } else {
          fn.input_col = layout.numColumns + static_cast<int32_t>(out->constants.size());
          fn.input_type = constant.typeKind;
          out->constants.push_back(std::move(constant));
        }
      } else {
        const auto& binding = bindings[channel];
        if (binding.column < 0) {
          return false;
        }
        fn.input_col = binding.column;
        if (raw) {
          if (!scalarKind(binding.type->kind())) {
            return false;
          }
          fn.input_type = static_cast<int32_t>(binding.type->kind());
        } else {
          // intermediate input: the result type follows the RAW input type (sum(REAL) returns REAL);
          // count's intermediate is a BIGINT whatever it counted (count(*) has no raw input at all)
          if (fn.kind == VX355_AGG_COUNT) {
            fn.input_type = VX355_BIGINT;
          } else if (aggregate.rawInputTypes.size() != 1 || !scalarKind(aggregate.rawInputTypes[0]->kind())) {
            return false;
          } else {
            fn.input_type = static_cast<int32_t>(aggregate.rawInputTypes[0]->kind());
          }
          if (fn.kind == VX355_AGG_AVG) {
            // ROW(DOUBLE sum, BIGINT count), flattened by DecodedBatch into two adjacent columns
            const auto& type = binding.type;
            if (!type->isRow() || type->size() != 2 || type->childAt(0)->kind() != TypeKind::DOUBLE ||
                type->childAt(1)->kind() != TypeKind::BIGINT) {
              return false;
            }
            fn.input_col2 = fn.input_col + 1;
          } else if (!scalarKind(binding.type->kind())) {
            return false;
          }
        }
      }
    }
    if (aggregate.mask) {
      const auto channel = exec::exprToChannel(aggregate.mask.get(), inputType);
      if (channel == kConstantChannel || bindings[channel].column < 0 ||
          bindings[channel].column >= VX355_PROJECTION_COL_BASE || bindings[channel].type->kind() != TypeKind::BOOLEAN) {
        return false;
      }
      fn.mask_col = bindings[channel].column;
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
  // HashAggregation promises no order of its groups: the reference lists RowContainer order
  // (exec/GroupingSet.cpp:828-839), which changes with the number of Drivers, with spilling and with
  // rehashes, and its own tests compare results as multisets (exec/tests/utils/QueryAssertions.h:36-42
  // MaterializedRowMultiset). Whatever consumes this operator in a plan - a final aggregation, an
  // exchange, an ORDER BY - does not depend on it, so the library may list groups in table order and
  // move records without row numbers through its radix passes (VX355_AGG_UNORDERED_OUTPUT:
  // 18.4 instead of 24.9 ms on BASELINE config 4).
  out->c.flags = VX355_AGG_UNORDERED_OUTPUT;
  return true;
}

bool toAggSpec(const core::AggregationNode& node, AggSpec* out) {
  const auto& inputType = node.sources()[0]->outputType();
  const ColumnLayout layout(inputType);
  return toAggSpec(node, identityBindings(inputType, layout), layout, out);
}

bool adaptDriver(const exec::DriverFactory& factory, exec::Driver& driver) {
  bool replaced = adaptJoins(factory, driver);  // HashBuild / HashProbe: Vx355JoinAdapter.cpp
  auto operators = driver.operators();
  for (int32_t i = 0; i < static_cast<int32_t>(operators.size()); ++i) {
    auto* aggregation = dynamic_cast<exec::HashAggregation*>(operators[i]);
    if (aggregation == nullptr) {
      continue;
    }
    auto node = planNodeOf<core::AggregationNode>(factory, aggregation->planNodeId());
    if (node == nullptr) {
      continue;
    }
    // FilterProject -> HashAggregation fusion: the operator in front is a FilterProject whose plan
    // nodes (FilterNode and / or ProjectNode, exec/LocalPlanner.cpp:517-533) feed this aggregation
    // and whose expressions the library evaluates itself; raw-input steps only.
    int32_t begin = i;
    AggSpec spec;
    FusedInput fused;
    bool isFused = false;
    auto* filterProject = i > 0 ? dynamic_cast<exec::FilterProject*>(operators[i - 1]) : nullptr;
    if (filterProject != nullptr && exec::isRawInput(node->step())) {
      auto project = planNodeOf<core::ProjectNode>(factory, filterProject->planNodeId());
      auto filter = planNodeOf<core::FilterNode>(
          factory, project != nullptr ? project->sources()[0]->id() : filterProject->planNodeId());
      const bool feeds = node->sources()[0]->id() == filterProject->planNodeId();
      if (feeds && (project != nullptr || filter != nullptr) && toFusedInput(filter.get(), project.get(), &fused) &&
          toAggSpec(*node, fused.bindings, ColumnLayout(fused.scanType), &spec)) {
        isFused = true;
        begin = i - 1;
      } else {
        spec = AggSpec{};
      }
    }
    if (!isFused && !toAggSpec(*node, &spec)) {
      continue;
    }
    vx355_agg* handle = nullptr;
    if (vx355_agg_create(&spec.c, &handle) != VX355_OK) {
      continue;  // VX355_EUNSUPPORTED: the CPU operator stays
    }
    if (isFused &&
        vx355_agg_set_fused_input(
            handle, fused.terms.data(), static_cast<int32_t>(fused.terms.size()), fused.projections.data(),
            static_cast<int32_t>(fused.projections.size())) != VX355_OK) {
      // the library declined this shape of fusion: aggregate behind the CPU FilterProject instead
      vx355_agg_destroy(handle);
      handle = nullptr;
      isFused = false;
      begin = i;
      spec = AggSpec{};
      if (!toAggSpec(*node, &spec) || vx355_agg_create(&spec.c, &handle) != VX355_OK) {
        continue;
      }
    }
    std::vector<std::unique_ptr<exec::Operator>> replacement;
    replacement.push_back(std::make_unique<Vx355HashAggregation>(
        operators[begin]->operatorId(), driver.driverCtx(), node, handle, spec.layout, std::move(spec.constants)));
    factory.replaceOperators(driver, begin, i + 1, std::move(replacement));
    replaced = true;
    operators = driver.operators();
    i = begin;
  }
  return replaced;
}

void registerVx355(int device, int64_t memoryLimitBytes) {
  VELOX_CHECK_EQ(vx355_init(device), VX355_OK, "{}", vx355_last_error());
  VELOX_CHECK_EQ(vx355_set_memory_limit(memoryLimitBytes), VX355_OK, "{}", vx355_last_error());
  exec::DriverFactory::registerAdapter(exec::DriverAdapter{"vx355", /*inspect=*/{}, adaptDriver});
}

// ---- batches in, columns out -------------------------------------------------------------------

void DecodedBatch::addChild(const BaseVector& child, vector_size_t /*numRows*/) {
  decoded_.push_back(std::make_unique<DecodedVector>(*child.loadedVector()));  // all rows
  auto& d = *decoded_.back();
  vx355_column col{};
  col.type_kind = static_cast<int32_t>(child.typeKind());
  col.mem = VX355_MEM_HOST;
  if (!scalarKind(child.typeKind())) {
    columns_.push_back(col);  // a channel nobody references (toAggSpec / toJoinPlan refused the others)
    return;
  }
  col.values = d.data<void>();
  col.nulls = d.nulls();  // bit per top-level row after decoding, 1 = valid: vx355's polarity
  if (d.isConstantMapping()) {
    col.encoding = VX355_CONSTANT;
    static const uint64_t kNull = 0;
    if (d.isNullAt(0)) {
      col.nulls = &kNull;
      col.values = nullptr;
    } else {
      // the single value sits at index(0) of the base (BOOLEAN values are bits: re-based below)
      col.nulls = nullptr;
      if (child.typeKind() == TypeKind::BOOLEAN) {
        static const uint64_t kTrue = 1, kFalse = 0;
        col.values = bits::isBitSet(d.data<uint64_t>(), d.index(0)) ? &kTrue : &kFalse;
      } else {
        col.values = d.data<char>() + static_cast<size_t>(d.index(0)) * child.type()->cppSizeInBytes();
      }
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

void DecodedBatch::addStruct(const VectorPtr& child, int32_t numFields, vector_size_t numRows) {
  // ROW(sum, count) from an upstream partial aggregation is a flat RowVector; anything wrapped
  // (a dictionary over structs after a local exchange) is flattened first
  VectorPtr flat = child;
  if (flat->encoding() != VectorEncoding::Simple::ROW) {
    BaseVector::flattenVector(flat);
    flattened_.push_back(flat);
  }
  auto* row = flat->as<RowVector>();
  VELOX_CHECK_NOT_NULL(row);
  VELOX_CHECK_EQ(static_cast<int32_t>(row->childrenSize()), numFields);
  const uint64_t* structNulls = row->rawNulls();
  for (int32_t f = 0; f < numFields; ++f) {
    addChild(*row->childAt(f), numRows);
    if (structNulls == nullptr) {
      continue;
    }
    // a field is null where the struct is: merge the two bitmaps (AND of the valid bits)
    auto& col = columns_.back();
    VELOX_CHECK(col.encoding == VX355_FLAT, "fields of a flat RowVector are read flat");
    mergedNulls_.emplace_back(structNulls, structNulls + bits::nwords(numRows));
    auto& merged = mergedNulls_.back();
    if (col.nulls != nullptr) {
      for (size_t w = 0; w < merged.size(); ++w) {
        merged[w] &= col.nulls[w];
      }
    }
    col.nulls = merged.data();
  }
}

DecodedBatch::DecodedBatch(const RowVector& input, const ColumnLayout& layout, const std::vector<ConstantColumn>& constants) {
  const auto numRows = input.size();
  VELOX_CHECK_EQ(layout.first.size(), input.childrenSize());
  columns_.reserve(static_cast<size_t>(layout.numColumns) + constants.size());
  for (size_t c = 0; c < input.childrenSize(); ++c) {
    const auto& child = input.childAt(static_cast<column_index_t>(c));
    if (layout.count[c] > 1 || structOfScalars(child->type())) {
      addStruct(child, layout.count[c], numRows);
    } else {
      addChild(*child, numRows);
    }
  }
  for (const auto& constant : constants) {
    vx355_column col{};
    col.type_kind = constant.typeKind;
    col.encoding = VX355_CONSTANT;
    col.mem = VX355_MEM_HOST;
    static const uint64_t kNull = 0;
    if (constant.isNull) {
      col.nulls = &kNull;
    } else if (!constant.longString.empty() || constant.typeKind == VX355_VARCHAR || constant.typeKind == VX355_VARBINARY) {
      // the StringView of a string constant points into the ConstantColumn the operator owns
      mergedNulls_.emplace_back(2, 0);
      StringView view = constant.longString.empty()
          ? *reinterpret_cast<const StringView*>(constant.value)
          : StringView(constant.longString.data(), constant.longString.size());
      std::memcpy(mergedNulls_.back().data(), &view, sizeof(view));
      col.values = mergedNulls_.back().data();
    } else {
      col.values = constant.value;
    }
    columns_.push_back(col);
  }
  batch_.num_rows = numRows;
  batch_.num_cols = static_cast<int32_t>(columns_.size());
  batch_.cols = columns_.data();
}

DecodedBatch::DecodedBatch(const RowVector& input)
    : DecodedBatch(input, ColumnLayout(asRowType(input.type())), {}) {}

OutColumns::OutColumns(RowVector& result) : result_(result) {
  auto add = [&](const VectorPtr& child) {
    vx355_out_column col{};
    col.type_kind = static_cast<int32_t>(child->typeKind());
    col.mem = VX355_MEM_HOST;
    col.values = child->values()->asMutable<void>();  // flat scalar vectors BaseVector::create made
    col.nulls = child->mutableRawNulls();
    columns_.push_back(col);
  };
  for (size_t i = 0; i < result.childrenSize(); ++i) {
    auto& child = result.childAt(static_cast<column_index_t>(i));
    if (child->type()->isRow()) {
      // ROW(DOUBLE sum, BIGINT count): the library writes the fields as two flat columns
      for (auto& field : child->as<RowVector>()->children()) {
        add(field);
      }
    } else {
      add(child);
    }
  }
}

void OutColumns::finish(vector_size_t numRows) {
  for (size_t i = 0; i < result_.childrenSize(); ++i) {
    auto& child = result_.childAt(static_cast<column_index_t>(i));
    if (!child->type()->isRow()) {
      continue;
    }
    auto* row = child->as<RowVector>();
    const uint64_t* fieldNulls = row->childAt(0)->rawNulls();
    if (fieldNulls == nullptr) {
      continue;
    }
    bool any = false;
    for (vector_size_t r = 0; r < numRows && !any; ++r) {
      any = bits::isBitNull(fieldNulls, r);
    }
    if (any) {
      std::memcpy(row->mutableRawNulls(), fieldNulls, bits::nwords(numRows) * sizeof(uint64_t));
    }
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
    vx355_agg* handle,
    ColumnLayout layout,
    std::vector<ConstantColumn> constants)
    : Operator(driverCtx, node->outputType(), operatorId, node->id(), "Vx355HashAggregation"),
      handle_(handle),
      layout_(std::move(layout)),
      constants_(std::move(constants)),
      isPartialOutput_(exec::isPartialOutput(node->step())),
      isGlobal_(node->groupingKeys().empty()),
      maxPartialMemory_(static_cast<int64_t>(driverCtx->queryConfig().maxPartialAggregationMemoryUsage())) {}

Vx355HashAggregation::~Vx355HashAggregation() {
  if (handle_ != nullptr) {
    vx355_agg_destroy(handle_);
  }
}

void Vx355HashAggregation::check(int status) {
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

exec::BlockingReason Vx355HashAggregation::isBlocked(ContinueFuture* future) {
  // The library takes whole chunks of ~1 M rows (parallel ingest): a few thousand vectors may be in
  // flight. Bound the memory they pin: beyond the bound the operator waits for the library's queue
  // (the GPU side is the bottleneck then; the queue drains in milliseconds).
  releaseCompleted();
  constexpr size_t kMaxInFlight = 4096;
  if (inFlight_.size() >= kMaxInFlight) {
    check(vx355_agg_wait(handle_));
    releaseCompleted();
  }
  // Output: the page is queued here, behind the batches and noMoreInput, and the Driver goes off the
  // thread until the worker has filled it (exec/Driver.cpp:538-800: blocked operators yield the thread).
  if (wantsOutput()) {
    if (page_ == nullptr) {
      startPage();
    }
    {
      std::lock_guard<std::mutex> lock(page_->mutex);
      if (!page_->done.load(std::memory_order_acquire)) {
        page_->promises.emplace_back("Vx355HashAggregation::getOutput");
        *future = page_->promises.back().getSemiFuture();
        return exec::BlockingReason::kWaitForConnector;  // (as TableScan waiting for an asynchronous source)
      }
    }
  }
  return exec::BlockingReason::kNotBlocked;
}

void Vx355HashAggregation::onPageDone(void* arg, int /*status*/, int32_t /*numRows*/, int32_t /*finished*/) {
  // on the library's worker thread: wake the Driver, nothing else (the status comes with the result)
  auto* page = static_cast<Page*>(arg);
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

void Vx355HashAggregation::startPage() {
  const auto maxRows = outputBatchRows();
  auto page = std::make_unique<Page>();
  page->result = std::static_pointer_cast<RowVector>(BaseVector::create(outputType_, maxRows, pool()));
  page->out = std::make_unique<OutColumns>(*page->result);  // (gives every column a nulls buffer)
  check(vx355_agg_get_output_async(handle_, page->out->data(), page->out->size(), maxRows, &onPageDone, page.get(),
                                   &page->ticket));
  page_ = std::move(page);
}

void Vx355HashAggregation::addInput(RowVectorPtr input) {
  auto decoded = std::make_unique<DecodedBatch>(*input, layout_, constants_);
  int64_t ticket = 0;
  check(vx355_agg_add_input_async(handle_, decoded->get(), &ticket));
  inFlight_.push_back(InFlight{ticket, std::move(input), std::move(decoded)});
  if (isPartialOutput_ && !isGlobal_ && partialFull()) {
    // HashAggregation.cpp:191-236,293-327: flush the partial groups, continue with an empty table
    check(vx355_agg_flush(handle_));
    flushing_ = true;
  }
}

bool Vx355HashAggregation::partialFull() {
  // vx355_agg_table_bytes never waits and never seals the open ingest chunk (vx355_agg_get_stats
  // drains the queue): the size is as of the last completed batch, so ask only when one completed
  int64_t submitted = 0, completed = 0;
  check(vx355_agg_poll(handle_, &submitted, &completed));
  if (completed == completedAtLastCheck_) {
    return false;
  }
  completedAtLastCheck_ = completed;
  // GroupingSet::isPartialFull (exec/GroupingSet.cpp:190-223) compares the bytes the groups occupy
  // with max_partial_aggregation_memory every time; the library reports them without the part of its
  // allocation that a flush left empty (vx355_agg_bytes_in_use).
  int64_t bytesInUse = 0;
  check(vx355_agg_bytes_in_use(handle_, &bytesInUse));
  return bytesInUse > maxPartialMemory_;
}

void Vx355HashAggregation::recordStats() {
  if (statsRecorded_ || handle_ == nullptr) {
    return;
  }
  statsRecorded_ = true;
  // HashAggregation::updateRuntimeStats -> BaseHashTable::addRuntimeStats (exec/HashAggregation.cpp:259-280,
  // exec/HashTable.cpp: the names of exec/HashTable.h:155-159), once, when the operator is done
  vx355_agg_stats table{};
  if (vx355_agg_get_stats(handle_, &table) == VX355_OK) {
    addRuntimeStat(exec::BaseHashTable::kCapacity, RuntimeCounter(table.capacity));
    addRuntimeStat(exec::BaseHashTable::kNumRehashes, RuntimeCounter(table.num_rehashes));
    addRuntimeStat(exec::BaseHashTable::kNumDistinct, RuntimeCounter(table.num_groups));
    addRuntimeStat(exec::BaseHashTable::kHashMode, RuntimeCounter(table.hash_mode));
    if (table.num_flushes > 0) {
      addRuntimeStat(exec::HashAggregation::kFlushTimes, RuntimeCounter(table.num_flushes));
    }
  }
  vx355_gpu_stats gpu{};
  if (vx355_agg_get_gpu_stats(handle_, &gpu) == VX355_OK) {
    recordGpuStats(*this, gpu, table.table_bytes);
  }
}

void recordGpuStats(exec::Operator& op, const vx355_gpu_stats& gpu, int64_t tableBytes) {
  // SURVEY.md section 5 "Metrics": the reference's names plus what only a GPU operator has
  op.addRuntimeStat("gpu.kernelNanos", RuntimeCounter(gpu.busy_nanos, RuntimeCounter::Unit::kNanos));
  op.addRuntimeStat("gpu.h2dBytes", RuntimeCounter(gpu.h2d_bytes, RuntimeCounter::Unit::kBytes));
  op.addRuntimeStat("gpu.d2hBytes", RuntimeCounter(gpu.d2h_bytes, RuntimeCounter::Unit::kBytes));
  op.addRuntimeStat("gpu.hbmBytesRead", RuntimeCounter(gpu.input_bytes, RuntimeCounter::Unit::kBytes));
  op.addRuntimeStat("gpu.hbmBytesWritten", RuntimeCounter(gpu.d2h_bytes + tableBytes, RuntimeCounter::Unit::kBytes));
  op.addRuntimeStat("gpu.kernelLaunches", RuntimeCounter(gpu.launches));
}

void Vx355HashAggregation::noMoreInput() {
  Operator::noMoreInput();
  check(vx355_agg_no_more_input_async(handle_, nullptr));  // queued behind the batches: nobody waits here
}

RowVectorPtr Vx355HashAggregation::getOutput() {
  if (!wantsOutput()) {
    return nullptr;
  }
  if (page_ == nullptr) {
    startPage();  // (a Driver that did not ask isBlocked() first)
  }
  if (!page_->done.load(std::memory_order_acquire)) {
    check(vx355_agg_wait(handle_));
    while (!page_->done.load(std::memory_order_acquire)) {
      std::this_thread::yield();  // (the callback runs right behind the ticket's completion)
    }
  }
  { std::lock_guard<std::mutex> callbackLeft(page_->mutex); }  // (the worker sets 'done' under this lock)
  releaseCompleted();
  auto page = std::move(page_);
  int32_t numRows = 0, finished = 0;
  check(vx355_agg_output_result(handle_, page->ticket, &numRows, &finished));
  if (finished) {
    if (flushing_) {
      flushing_ = false;  // the table is empty again: GroupingSet::resetTable
    } else {
      finished_ = true;
      recordStats();
    }
  }
  if (numRows == 0) {
    return nullptr;
  }
  page->out->finish(numRows);
  auto result = std::move(page->result);
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
  recordStats();
  if (handle_ != nullptr) {
    vx355_agg_destroy(handle_);
    handle_ = nullptr;
  }
  inFlight_.clear();
  page_.reset();  // (the handle's worker is gone: no callback can come)
  Operator::close();
}

}  // namespace facebook::velox::vx355
