// TEST INFRASTRUCTURE ONLY — CPU oracle, see oracle.h.
// GroupingSet / HashAggregation / HashBuild / HashProbe / filter compaction /
// partition function restated on top of table.h, and the C API.
#include "oracle.h"

#include <type_traits>
#include <cmath>
#include <map>
#include <set>
#include <string>
#include <cstdio>

#include "presto_page.h"
#include "table.h"

namespace orc {
namespace {

thread_local std::string gLastError;
static int gJoinBuildThreads = 1;

bool isIntKind(int32_t k) { return k >= VX355_BOOLEAN && k <= VX355_BIGINT; }
bool isStringKind(int32_t k) { return k == VX355_VARCHAR || k == VX355_VARBINARY; }

// ---- aggregate functions (Appendix C of SURVEY.md) ------------------------

// Accumulator bytes: sum/count/min/max 8, avg 16 {double sum; int64 count}
// (AverageAggregateBase.h:66-69).
int32_t accBytes(const vx355_agg_fn& f) { return f.kind == VX355_AGG_AVG ? 16 : 8; }

bool rawInput(int32_t step) { return step == VX355_STEP_PARTIAL || step == VX355_STEP_SINGLE; }
bool finalOutput(int32_t step) { return step == VX355_STEP_FINAL || step == VX355_STEP_SINGLE; }

// Intermediate / final result type of an aggregate (SumAggregate.cpp:39-118,
// CountAggregate.cpp, MinMaxAggregateBase.cpp:34-99, AverageAggregate.cpp).
void outputTypes(const vx355_agg_fn& f, int32_t step, std::vector<int32_t>& out) {
  bool fin = finalOutput(step);
  switch (f.kind) {
    case VX355_AGG_SUM:
      if (isIntKind(f.input_type)) {
        out.push_back(VX355_BIGINT);
      } else if (f.input_type == VX355_REAL) {
        out.push_back(fin ? VX355_REAL : VX355_DOUBLE);
      } else {
        out.push_back(VX355_DOUBLE);
      }
      break;
    case VX355_AGG_COUNT:
    case VX355_AGG_COUNT_STAR:
      out.push_back(VX355_BIGINT);
      break;
    case VX355_AGG_MIN:
    case VX355_AGG_MAX:
      out.push_back(f.input_type);
      break;
    case VX355_AGG_AVG:
      if (fin) {
        out.push_back(f.input_type == VX355_REAL ? VX355_REAL : VX355_DOUBLE);
      } else {
        out.push_back(VX355_DOUBLE);
        out.push_back(VX355_BIGINT);
      }
      break;
  }
}

// Which overflow rule sum(BIGINT) follows (orc_set_sum_overflow_rule):
//   0 = the reference's: every add is a checkedPlus on the running sum, in input order
//       (vector/AggregationHook.h:126-135, SumAggregate.cpp:24): a PREFIX that leaves int64 throws;
//   1 = "total": the exact (128-bit) total is checked once, when it is read out — the one rule a
//       parallel implementation can follow whatever order it adds in (libvx355).
// The two agree unless a prefix overflows and later rows bring the sum back into range.
int gSumOverflowRule = 0;

// checkedPlus (vector/AggregationHook.h:126-135 SumHook::add with
// Overflow=false, SumAggregate.cpp:24).
int64_t checkedPlus(int64_t a, int64_t b) {
  int64_t r;
  if (__builtin_add_overflow(a, b, &r)) {
    throw UserError("integer overflow: " + std::to_string(a) + " + " + std::to_string(b));
  }
  return r;
}

// Comparison of two values the way the reference's comparison functions do it
// (functions/prestosql/Comparisons.h:42-121 -> util::floating_point::NaNAware*): for REAL / DOUBLE
// NaN equals NaN and is greater than every other value; integers compare as usual.
template <typename T>
bool compareLikeVelox(int32_t cmp, T a, T b) {
  if constexpr (std::is_floating_point<T>::value) {
    const bool an = std::isnan(a), bn = std::isnan(b);
    const bool eq = (an && bn) || a == b;
    const bool lt = !an && (bn || a < b);
    const bool gt = !bn && (an || a > b);
    switch (cmp) {
      case VX355_CMP_EQ:
        return eq;
      case VX355_CMP_NE:
        return !eq;
      case VX355_CMP_LT:
        return lt;
      case VX355_CMP_LE:
        return lt || eq;
      case VX355_CMP_GT:
        return gt;
      default:
        return gt || eq;
    }
  } else {
    switch (cmp) {
      case VX355_CMP_EQ:
        return a == b;
      case VX355_CMP_NE:
        return a != b;
      case VX355_CMP_LT:
        return a < b;
      case VX355_CMP_LE:
        return a <= b;
      case VX355_CMP_GT:
        return a > b;
      default:
        return a >= b;
    }
  }
}

// NaN-aware compare (MinMaxAggregateBase.cpp:174-184, :267-277): NaN is larger
// than everything.
bool lessThan(double a, double b) {
  if (std::isnan(b)) {
    return !std::isnan(a);
  }
  if (std::isnan(a)) {
    return false;
  }
  return a < b;
}
bool greaterThan(double a, double b) { return lessThan(b, a); }

void writeOut(vx355_out_column& col, int32_t row, bool isNull, const void* value, int width) {
  if (col.nulls) {
    setBit(col.nulls, row, !isNull);
  }
  if (isNull) {
    if (col.type_kind == VX355_BOOLEAN) {
      setBit(static_cast<uint64_t*>(col.values), row, false);
    } else {
      std::memset(static_cast<char*>(col.values) + static_cast<int64_t>(row) * width, 0, width);
    }
    return;
  }
  if (col.type_kind == VX355_BOOLEAN) {
    setBit(static_cast<uint64_t*>(col.values), row, *static_cast<const uint8_t*>(value) != 0);
    return;
  }
  std::memcpy(static_cast<char*>(col.values) + static_cast<int64_t>(row) * width, value, width);
}

// Stored 8-byte key slot back to the column type (RowContainer::extractColumn,
// RowContainer.h:439-536).
void extractStored(const char* row, const RowContainer::Column& c, vx355_out_column& out,
                   int32_t outRow) {
  bool isNull = row[c.nullOffset] != 0;
  const char* p = row + c.offset;
  switch (c.kind) {
    case VX355_BOOLEAN: {
      int64_t v;
      std::memcpy(&v, p, 8);
      uint8_t b = v != 0;
      writeOut(out, outRow, isNull, &b, 0);
      break;
    }
    case VX355_TINYINT:
    case VX355_SMALLINT:
    case VX355_INTEGER:
    case VX355_BIGINT: {
      int64_t v;
      std::memcpy(&v, p, 8);
      writeOut(out, outRow, isNull, &v, kindWidth(c.kind));  // little endian truncation
      break;
    }
    case VX355_REAL:
      writeOut(out, outRow, isNull, p, 4);
      break;
    case VX355_DOUBLE:
      writeOut(out, outRow, isNull, p, 8);
      break;
    default:
      writeOut(out, outRow, isNull, p, 16);
  }
}

}  // namespace

// exec/GroupingSet.{h,cpp} + exec/HashAggregation.{h,cpp}.
class Aggregation {
  // rule 1: exact totals of the BIGINT sums, by accumulator address
  std::unordered_map<char*, __int128> wideSums_;
  void addBigintSum(char* a, int64_t v) {
    auto* s = reinterpret_cast<int64_t*>(a);
    if (gSumOverflowRule == 0) {
      *s = checkedPlus(*s, v);
      return;
    }
    __int128& w = wideSums_[a];
    w += v;
    *s = static_cast<int64_t>(static_cast<uint64_t>(*s) + static_cast<uint64_t>(v));
  }
  void checkBigintTotal(char* a) {
    if (gSumOverflowRule == 0) {
      return;
    }
    auto it = wideSums_.find(a);
    if (it != wideSums_.end() && (it->second > INT64_MAX || it->second < INT64_MIN)) {
      throw UserError("integer overflow: the total of sum(BIGINT) does not fit int64");
    }
  }

 public:
  Aggregation(const vx355_agg_spec& spec, bool hashAdaptivity)
      : step_(spec.step), ignoreNullKeys_(spec.ignore_null_keys != 0) {
    keyCols_.assign(spec.key_cols, spec.key_cols + spec.num_keys);
    keyKinds_.assign(spec.key_types, spec.key_types + spec.num_keys);
    aggs_.assign(spec.aggs, spec.aggs + spec.num_aggs);
    std::vector<int32_t> bytes;
    for (auto& f : aggs_) {
      bytes.push_back(accBytes(f));
      if ((f.flags & VX355_AGG_FN_DISTINCT) && step_ != VX355_STEP_SINGLE) {
        // GroupingSet.cpp:117-121 (isPartial_), and no plan feeds intermediate input to one
        throw UserError("Partial aggregations over distinct inputs are not supported");
      }
    }
    table_ = std::make_unique<HashTable>(keyKinds_, bytes, std::vector<int32_t>{}, false, false,
                                         ignoreNullKeys_);
    if (!hashAdaptivity && !keyKinds_.empty()) {
      table_->forceGenericHashMode();
    }
    if (keyKinds_.empty()) {
      // Global aggregation: one row, always (GroupingSet.cpp:623-669).
      globalRow_ = table_->rows()->newRow();
      for (size_t i = 0; i < aggs_.size(); ++i) {
        initGroup(globalRow_, i);
      }
    }
  }

  // GroupingSet::addInput (GroupingSet.cpp:190-223) -> addInputForActiveRows
  // (:288-365).
  void addInput(const vx355_batch& batch) {
    const int32_t n = batch.num_rows;
    inputRows_ += n;
    if (n == 0) {
      return;
    }
    std::vector<uint64_t> rows((n + 63) / 64, ~0ULL);
    if (n & 63) {
      rows.back() = (1ULL << (n & 63)) - 1;
    }
    std::vector<char*> groupsStorage;
    char** groups;
    HashLookup lookup;
    if (keyKinds_.empty()) {
      groupsStorage.assign(n, globalRow_);
      groups = groupsStorage.data();
    } else {
      std::vector<Decoded> keys;
      for (auto c : keyCols_) {
        keys.emplace_back(&batch.cols[c]);
      }
      table_->prepareForGroupProbe(lookup, keys, n, rows);
      table_->groupProbe(lookup, keys);
      groups = lookup.hits.data();
      // Aggregate::initializeNewGroups (Aggregate.h:152).
      for (auto r : lookup.newGroups) {
        for (size_t i = 0; i < aggs_.size(); ++i) {
          initGroup(groups[r], i);
        }
      }
    }
    for (size_t i = 0; i < aggs_.size(); ++i) {
      update(i, batch, rows.data(), n, groups);
    }
  }

  void noMoreInput() { noMoreInput_ = true; }

  std::vector<int32_t> outputKinds() const {
    std::vector<int32_t> out = keyKinds_;
    for (auto& f : aggs_) {
      outputTypes(f, step_, out);
    }
    return out;
  }

  // GroupingSet::getOutput (GroupingSet.cpp:810-841) + extractGroups (:843-884):
  // rows in RowContainer order == first-seen order of groups.
  void getOutput(vx355_out_column* cols, int32_t numCols, int32_t maxRows, int32_t* nOut,
                 int32_t* finished) {
    auto kinds = outputKinds();
    if (numCols != static_cast<int32_t>(kinds.size())) {
      throw std::runtime_error("getOutput: wrong number of output columns");
    }
    applyDistinctSets();
    const auto& rows = table_->rows()->rows();
    int64_t total = static_cast<int64_t>(rows.size());
    int32_t n = static_cast<int32_t>(std::min<int64_t>(maxRows, total - outputCursor_));
    for (int32_t i = 0; i < n; ++i) {
      char* row = rows[outputCursor_ + i];
      int c = 0;
      for (size_t k = 0; k < keyKinds_.size(); ++k) {
        extractStored(row, table_->rows()->keys()[k], cols[c++], i);
      }
      for (size_t a = 0; a < aggs_.size(); ++a) {
        extract(a, row, cols, c, i);
      }
    }
    outputCursor_ += n;
    *nOut = n;
    *finished = outputCursor_ >= total;
  }

  void stats(vx355_agg_stats* out) const {
    out->num_groups = table_->rows()->numRows();
    out->capacity = static_cast<int64_t>(table_->capacity());
    out->num_rehashes = table_->numRehashes();
    out->hash_mode = static_cast<int32_t>(table_->hashMode());
    out->reserved = 0;
    out->input_rows = inputRows_;
    out->deferred_rows = 0;
  }

 private:
  char* acc(char* group, size_t i) const { return group + table_->rows()->accOffset(i); }
  char& accNull(char* group, size_t i) const { return group[table_->rows()->accNullOffset(i)]; }

  static bool usesDistinctSet(const vx355_agg_fn& f) {
    // min / max over the set of values == over the values; they go the plain way here.
    return (f.flags & VX355_AGG_FN_DISTINCT) &&
        (f.kind == VX355_AGG_SUM || f.kind == VX355_AGG_COUNT || f.kind == VX355_AGG_AVG);
  }

  // TypedDistinctAggregations::extractValues (DistinctAggregations.cpp:240-285): each group's
  // distinct values go through addSingleGroupRawInput in the set's insertion order.
  void applyDistinctSets() {
    for (auto& entry : distinct_) {
      char* g = entry.first.first;
      const size_t i = entry.first.second;
      const auto& f = aggs_[i];
      char* a = acc(g, i);
      for (const SetValue& v : entry.second.ordered) {
        if (v.isNull) {
          continue;
        }
        const bool isInt = isIntKind(f.input_type);
        switch (f.kind) {
          case VX355_AGG_COUNT:
            ++*reinterpret_cast<int64_t*>(a);
            break;
          case VX355_AGG_SUM:
            accNull(g, i) = 0;
            if (isInt) {
              addBigintSum(a, v.i);
            } else {
              *reinterpret_cast<double*>(a) += v.d;
            }
            break;
          case VX355_AGG_AVG:
            accNull(g, i) = 0;
            *reinterpret_cast<double*>(a) += isInt ? static_cast<double>(v.i) : v.d;
            *reinterpret_cast<int64_t*>(a + 8) = checkedPlus(*reinterpret_cast<int64_t*>(a + 8), 1);
            break;
        }
      }
    }
    distinct_.clear();
  }

  // initializeNewGroups: sum/min/max/avg start null (SumAggregateBase.h:144-151,
  // MinMaxAggregateBase.cpp:191-201/:293-303, AverageAggregateBase.h), count
  // starts at 0 and is never null (CountAggregate.cpp:135-142).
  void initGroup(char* group, size_t i) {
    const auto& f = aggs_[i];
    accNull(group, i) = (f.kind == VX355_AGG_COUNT || f.kind == VX355_AGG_COUNT_STAR) ? 0 : 1;
    char* a = acc(group, i);
    if (f.kind == VX355_AGG_MIN || f.kind == VX355_AGG_MAX) {
      bool isMin = f.kind == VX355_AGG_MIN;
      if (isIntKind(f.input_type)) {
        int64_t v = isMin ? limitMax(f.input_type) : limitMin(f.input_type);
        std::memcpy(a, &v, 8);
      } else {
        // MinMaxAggregateBase.cpp:293-303: min starts at quiet_NaN ("NaN is
        // considered larger than infinity"), :191-201: max at -infinity.
        double v = isMin ? std::numeric_limits<double>::quiet_NaN()
                         : -std::numeric_limits<double>::infinity();
        std::memcpy(a, &v, 8);
      }
    } else {
      std::memset(a, 0, accBytes(f));
    }
  }
  static int64_t limitMax(int32_t kind) {
    switch (kind) {
      case VX355_BOOLEAN:
        return 1;
      case VX355_TINYINT:
        return INT8_MAX;
      case VX355_SMALLINT:
        return INT16_MAX;
      case VX355_INTEGER:
        return INT32_MAX;
      default:
        return INT64_MAX;
    }
  }
  static int64_t limitMin(int32_t kind) {
    switch (kind) {
      case VX355_BOOLEAN:
        return 0;
      case VX355_TINYINT:
        return INT8_MIN;
      case VX355_SMALLINT:
        return INT16_MIN;
      case VX355_INTEGER:
        return INT32_MIN;
      default:
        return INT64_MIN;
    }
  }

  // Aggregate::addRawInput (Aggregate.h:179) / addIntermediateResults (:227):
  // one pass over groups[] per aggregate, like the reference
  // (SimpleNumericAggregate.h:94 updateGroups).
  void update(size_t i, const vx355_batch& batch, const uint64_t* rows, int32_t n, char** groups) {
    const auto& f = aggs_[i];
    const bool raw = rawInput(step_);
    std::unique_ptr<Decoded> mask, in, in2;
    if (f.mask_col >= 0) {
      mask = std::make_unique<Decoded>(&batch.cols[f.mask_col]);
    }
    if (f.input_col >= 0) {
      in = std::make_unique<Decoded>(&batch.cols[f.input_col]);
    }
    if (f.input_col2 >= 0) {
      in2 = std::make_unique<Decoded>(&batch.cols[f.input_col2]);
    }
    const bool intSum = isIntKind(f.input_type);
    for (int32_t r = 0; r < n; ++r) {
      if (!bitSet(rows, r)) {
        continue;
      }
      if (mask && (mask->isNull(r) || !mask->int64At(r))) {
        continue;  // AggregationMasks: false or null mask excludes the row
      }
      char* g = groups[r];
      char* a = acc(g, i);
      if (usesDistinctSet(f)) {
        // TypedDistinctAggregations::addInput (DistinctAggregations.cpp:213-227):
        // SetAccumulator::addValue keeps each value, null included, once, in arrival order.
        DistinctSet& set = distinct_[{g, i}];
        SetValue v;
        v.isNull = in->isNull(r);
        std::string identity;
        if (!v.isNull) {
          if (isStringKind(f.input_type)) {
            // count(DISTINCT s): the set compares contents (SetAccumulator<StringView>)
            uint8_t tmp;
            const auto* sv = static_cast<const StringView*>(in->valuePtr(r, &tmp));
            identity.assign(sv->data(), sv->size);
          } else if (intSum) {
            v.i = in->int64At(r);
            v.image = static_cast<uint64_t>(v.i);
          } else {
            v.d = in->doubleAt(r);
            // NaNAwareEquals / NaNAwareHash (type/FloatingPointUtil.h): every NaN is one value,
            // and 0.0 == -0.0
            if (std::isnan(v.d)) {
              v.image = 0x7ff8000000000000ULL;
            } else if (v.d == 0.0) {
              v.image = 0;
            } else {
              std::memcpy(&v.image, &v.d, 8);
            }
          }
          if (!isStringKind(f.input_type)) {
            identity.assign(reinterpret_cast<const char*>(&v.image), 8);
          }
        }
        if (set.seen.insert({v.isNull, identity}).second) {
          set.ordered.push_back(v);
        }
        continue;
      }
      if (isStringKind(f.input_type) && (f.kind == VX355_AGG_MIN || f.kind == VX355_AGG_MAX)) {
        // MinMaxAggregateBase.cpp:395-419 (non-numeric doUpdate): SingleValueAccumulator holds
        // the current extreme; compare() is StringView::compare (bytes, then length). Raw and
        // intermediate input are the same thing (:377-381).
        if (in->isNull(r)) {
          continue;
        }
        uint8_t tmp;
        const auto* sv = static_cast<const StringView*>(in->valuePtr(r, &tmp));
        std::string v(sv->data(), sv->size);
        auto it = stringAcc_.find({g, i});
        if (it == stringAcc_.end()) {
          stringAcc_.emplace(std::make_pair(g, i), std::move(v));
        } else if (f.kind == VX355_AGG_MIN ? v < it->second : v > it->second) {
          it->second = std::move(v);
        }
        continue;
      }
      switch (f.kind) {
        case VX355_AGG_COUNT_STAR:
          if (raw) {
            ++*reinterpret_cast<int64_t*>(a);
          } else if (!in->isNull(r)) {
            *reinterpret_cast<int64_t*>(a) += in->int64At(r);  // CountAggregate.cpp:72-80
          }
          break;
        case VX355_AGG_COUNT:
          if (in->isNull(r)) {
            break;
          }
          if (raw) {
            ++*reinterpret_cast<int64_t*>(a);
          } else {
            *reinterpret_cast<int64_t*>(a) += in->int64At(r);
          }
          break;
        case VX355_AGG_SUM:
          if (in->isNull(r)) {
            break;
          }
          accNull(g, i) = 0;
          if (intSum) {
            addBigintSum(a, in->int64At(r));
          } else {
            *reinterpret_cast<double*>(a) += in->doubleAt(r);
          }
          break;
        case VX355_AGG_MIN:
        case VX355_AGG_MAX: {
          if (in->isNull(r)) {
            break;
          }
          accNull(g, i) = 0;
          bool isMin = f.kind == VX355_AGG_MIN;
          if (intSum) {
            auto* s = reinterpret_cast<int64_t*>(a);
            int64_t v = in->int64At(r);
            if (isMin ? v < *s : v > *s) {
              *s = v;
            }
          } else {
            auto* s = reinterpret_cast<double*>(a);
            double v = in->doubleAt(r);
            if (isMin ? lessThan(v, *s) : greaterThan(v, *s)) {
              *s = v;
            }
          }
          break;
        }
        case VX355_AGG_AVG: {
          auto* sum = reinterpret_cast<double*>(a);
          auto* cnt = reinterpret_cast<int64_t*>(a + 8);
          if (raw) {
            if (in->isNull(r)) {
              break;
            }
            accNull(g, i) = 0;
            *sum += in->doubleAt(r);  // AverageAggregateBase.h:241-258
            *cnt = checkedPlus(*cnt, 1);
          } else {
            if (in->isNull(r)) {
              break;
            }
            accNull(g, i) = 0;
            *sum += in->doubleAt(r);  // :265-330
            *cnt = checkedPlus(*cnt, in2->int64At(r));
          }
          break;
        }
      }
    }
  }

  // extractValues (Aggregate.h:277) / extractAccumulators (:289).
  void extract(size_t i, char* group, vx355_out_column* cols, int& c, int32_t outRow) {
    const auto& f = aggs_[i];
    const bool fin = finalOutput(step_);
    bool isNull = accNull(group, i) != 0;
    char* a = acc(group, i);
    switch (f.kind) {
      case VX355_AGG_COUNT:
      case VX355_AGG_COUNT_STAR:
        writeOut(cols[c++], outRow, false, a, 8);
        break;
      case VX355_AGG_SUM:
        if (isIntKind(f.input_type)) {
          checkBigintTotal(a);
          writeOut(cols[c++], outRow, isNull, a, 8);
        } else if (f.input_type == VX355_REAL && fin) {
          float v = static_cast<float>(*reinterpret_cast<double*>(a));
          writeOut(cols[c++], outRow, isNull, &v, 4);
        } else {
          writeOut(cols[c++], outRow, isNull, a, 8);
        }
        break;
      case VX355_AGG_MIN:
      case VX355_AGG_MAX:
        if (isStringKind(f.input_type)) {
          // extractValues (MinMaxAggregateBase.cpp:352-375): null without a value
          auto it = stringAcc_.find({group, i});
          StringView view{};
          if (it != stringAcc_.end()) {
            const std::string& v = it->second;
            view.size = static_cast<uint32_t>(v.size());
            if (view.isInline()) {
              std::memcpy(view.prefix, v.data(), v.size());
            } else {
              std::memcpy(view.prefix, v.data(), 4);
              view.value.data = v.data();
            }
          }
          writeOut(cols[c++], outRow, it == stringAcc_.end(), &view, 16);
          break;
        }
        if (isIntKind(f.input_type)) {
          if (f.input_type == VX355_BOOLEAN) {
            uint8_t b = *reinterpret_cast<int64_t*>(a) != 0;
            writeOut(cols[c++], outRow, isNull, &b, 0);
          } else {
            writeOut(cols[c++], outRow, isNull, a, kindWidth(f.input_type));
          }
        } else if (f.input_type == VX355_REAL) {
          float v = static_cast<float>(*reinterpret_cast<double*>(a));
          writeOut(cols[c++], outRow, isNull, &v, 4);
        } else {
          writeOut(cols[c++], outRow, isNull, a, 8);
        }
        break;
      case VX355_AGG_AVG: {
        double sum = *reinterpret_cast<double*>(a);
        int64_t cnt = *reinterpret_cast<int64_t*>(a + 8);
        if (fin) {
          // AverageAggregateBase.h:86-109
          if (f.input_type == VX355_REAL) {
            float v = isNull ? 0.f : static_cast<float>(sum / cnt);
            writeOut(cols[c++], outRow, isNull, &v, 4);
          } else {
            double v = isNull ? 0. : sum / cnt;
            writeOut(cols[c++], outRow, isNull, &v, 8);
          }
        } else {
          writeOut(cols[c++], outRow, isNull, &sum, 8);
          writeOut(cols[c++], outRow, isNull, &cnt, 8);
        }
        break;
      }
    }
  }

  int32_t step_;
  bool ignoreNullKeys_;
  std::vector<int32_t> keyCols_, keyKinds_;
  std::vector<vx355_agg_fn> aggs_;
  // aggregate::prestosql::SetAccumulator<T> per (group, DISTINCT aggregate)
  struct SetValue {
    bool isNull = false;
    int64_t i = 0;
    double d = 0;
    uint64_t image = 0;
  };
  struct DistinctSet {
    std::set<std::pair<bool, std::string>> seen;
    std::vector<SetValue> ordered;
  };
  std::map<std::pair<char*, size_t>, DistinctSet> distinct_;
  // SingleValueAccumulator of min / max over VARCHAR / VARBINARY, per (group, aggregate)
  std::map<std::pair<char*, size_t>, std::string> stringAcc_;
  std::unique_ptr<HashTable> table_;
  char* globalRow_ = nullptr;
  bool noMoreInput_ = false;
  int64_t outputCursor_ = 0;
  int64_t inputRows_ = 0;
};

// exec/HashBuild.{h,cpp}: one per build Driver.
class JoinBuild {
 public:
  explicit JoinBuild(const vx355_join_build_spec& spec)
      : joinType_(spec.join_type), nullAsValue_(spec.null_as_value != 0), nullAware_(spec.null_aware != 0) {
    keyCols_.assign(spec.key_cols, spec.key_cols + spec.num_keys);
    keyKinds_.assign(spec.key_types, spec.key_types + spec.num_keys);
    depCols_.assign(spec.dependent_cols, spec.dependent_cols + spec.num_dependents);
    depKinds_.assign(spec.dependent_types, spec.dependent_types + spec.num_dependents);
    table_ = std::make_unique<HashTable>(keyKinds_, std::vector<int32_t>{}, depKinds_, true, true,
                                         true);
  }

  // HashBuild::addInput (HashBuild.cpp:442-598).
  void addInput(const vx355_batch& batch) {
    const int32_t n = batch.num_rows;
    if (n == 0) {
      return;
    }
    std::vector<Decoded> keys, deps;
    for (auto c : keyCols_) {
      keys.emplace_back(&batch.cols[c]);
    }
    for (auto c : depCols_) {
      deps.emplace_back(&batch.cols[c]);
    }
    std::vector<uint64_t> rows((n + 63) / 64, ~0ULL);
    if (n & 63) {
      rows.back() = (1ULL << (n & 63)) - 1;
    }
    // Null keys never match: drop them, except for right / full joins whose
    // build rows all reach the output (HashBuild.cpp:475-494).
    // (HashBuild.cpp:257-268: right, full, right semi project and right anti retain null keys)
    // ... and so does nullAsValue (HashBuild.cpp:273,477): there the rows are ordinary table entries
    // ... and a null-aware join: with an extra filter its null-key build rows take part in the
    // result (HashBuild.cpp:257-268 keeps them unless "anti join, null aware, no filter";
    // HashProbe::evalFilterForNullAwareJoin, HashProbe.cpp:1639-1700, lists them)
    const bool keepNullKeys = joinType_ == VX355_JOIN_RIGHT || joinType_ == VX355_JOIN_FULL ||
        joinType_ == VX355_JOIN_RIGHT_SEMI_PROJECT || joinType_ == VX355_JOIN_RIGHT_ANTI || nullAsValue_ || nullAware_;
    for (auto& d : keys) {
      for (int32_t r = 0; r < n; ++r) {
        if (d.isNull(r)) {
          if (bitSet(rows.data(), r)) {
            hasNullKeys_ = true;
          }
          if (!keepNullKeys) {
            setBit(rows.data(), r, false);
          }
        }
      }
    }
    table_->analyzeJoinKeys(keys, n, rows.data());
    for (int32_t r = 0; r < n; ++r) {
      if (bitSet(rows.data(), r)) {
        table_->appendJoinRow(keys, deps, r);
      }
    }
  }

  std::unique_ptr<HashTable> table_;
  std::vector<int32_t> keyCols_, keyKinds_, depCols_, depKinds_;
  int32_t joinType_;
  bool nullAsValue_ = false;
  bool nullAware_ = false;
  bool hasNullKeys_ = false;
};

// The finished table handed over the HashJoinBridge.
struct JoinTable {
  std::unique_ptr<HashTable> table;
  std::vector<std::unique_ptr<HashTable>> others;
  // Row id = position in [table rows..., others' rows...] (the order
  // vx355_join_build_finish documents).
  std::vector<int64_t> containerBase;
  std::vector<RowContainer*> containers;
  std::vector<int32_t> depKinds;
  int64_t numRows = 0;
  bool hasNullKeys = false;
  // RowContainer probed flags (RowContainer.h probedFlagOffset_), by row id; set by every
  // probe of the table, read by the last one (HashProbe::getBuildSideOutput).
  std::vector<uint8_t> probed;
  // Counting joins (HashBuild.cpp:534-548, RowContainer::count): occurrences of each distinct
  // key not yet consumed by a probe row, by chain head; filled on first use.
  std::map<char*, int64_t> remaining;

  char* rowById(int64_t id) const {
    size_t c = containers.size() - 1;
    while (c > 0 && containerBase[c] > id) {
      --c;
    }
    return containers[c]->rows()[id - containerBase[c]];
  }
};

// exec/HashProbe.{h,cpp}.
class JoinProbe {
 public:
  JoinProbe(JoinTable* t, const vx355_join_probe_spec& spec)
      : table_(t), joinType_(spec.join_type), nullAware_(spec.null_aware != 0),
        nullAsValue_(spec.null_as_value != 0) {
    keyCols_.assign(spec.key_cols, spec.key_cols + spec.num_keys);
  }

  // HashProbe::addInput (HashProbe.cpp:796-900).
  void addInput(const vx355_batch& batch) {
    numRows_ = batch.num_rows;
    batch_ = &batch;
    cursorRow_ = 0;
    cursorChain_ = nullptr;
    chainOpen_ = false;
    keys_.clear();
    for (auto c : keyCols_) {
      keys_.emplace_back(&batch.cols[c]);
    }
    std::vector<uint64_t> rows((numRows_ + 63) / 64, ~0ULL);
    if (numRows_ & 63) {
      rows.back() = (1ULL << (numRows_ & 63)) - 1;
    }
    if (numRows_ == 0) {
      lookup_.reset(0);
      return;
    }
    table_->table->prepareForJoinProbe(lookup_, keys_, numRows_, rows, nullAsValue_);
    table_->table->joinProbe(lookup_, keys_);
  }

  // HashJoinNode::filter as a conjunction of vx355_join_filter_term (include/vx355.h).
  void setFilter(const vx355_join_filter_term* terms, int32_t n) { filter_.assign(terms, terms + n); }

  // HashTable::listJoinResults (HashTable.cpp:2133-2350) + HashProbe::evalFilter and the
  // per-join-kind bookkeeping around it (HashProbe.cpp:1276-1437,1487-1711), row at a time:
  // pairs in ascending probe-row order, all matches of one probe row contiguous, resumable.
  void getOutput(int32_t maxRows, int32_t* mapping, int32_t* buildRows, vx355_out_column* cols,
                 const int32_t* colIds, int32_t numCols, int32_t* nOut, int32_t* finished) {
    int32_t n = 0;
    const bool includeMisses = joinType_ == VX355_JOIN_LEFT || joinType_ == VX355_JOIN_FULL;
    const bool marksProbed = joinType_ == VX355_JOIN_RIGHT || joinType_ == VX355_JOIN_FULL;
    const bool buildSideOnly = joinType_ == VX355_JOIN_RIGHT_SEMI_FILTER ||
        joinType_ == VX355_JOIN_RIGHT_SEMI_PROJECT || joinType_ == VX355_JOIN_RIGHT_ANTI;
    while (cursorRow_ < numRows_ && n < maxRows) {
      char* hit = lookup_.hits[cursorRow_];
      if (buildSideOnly) {
        // processRightSemiNoFilter and friends: no probe-side output, only the probed flags
        // of the build rows a probe row matches (with the filter: of the passing pairs).
        for (char* cur = hit; cur; cur = table_->table->nextRow(cur)) {
          if (passes(cursorRow_, cur)) {
            table_->probed[rowId(cur)] = 1;
          }
        }
        ++cursorRow_;
        continue;
      }
      if (joinType_ == VX355_JOIN_COUNTING_LEFT_SEMI_FILTER || joinType_ == VX355_JOIN_COUNTING_ANTI) {
        // HashProbe.cpp:1345-1365: a match consumes one occurrence of the key while any is left.
        bool consumed = false;
        if (hit) {
          auto it = table_->remaining.find(hit);
          if (it == table_->remaining.end()) {
            int64_t count = 0;
            for (char* cur = hit; cur; cur = table_->table->nextRow(cur)) {
              ++count;
            }
            it = table_->remaining.emplace(hit, count).first;
          }
          if (it->second > 0) {
            --it->second;
            consumed = true;
          }
        }
        if (consumed == (joinType_ == VX355_JOIN_COUNTING_LEFT_SEMI_FILTER)) {
          emit(n++, cursorRow_, nullptr, mapping, buildRows, cols, colIds, numCols);
        }
        ++cursorRow_;
        continue;
      }
      if (nullAware_ && !filter_.empty() &&
          (joinType_ == VX355_JOIN_ANTI || joinType_ == VX355_JOIN_LEFT_SEMI_PROJECT)) {
        // HashProbe::evalFilter + evalFilterForNullAwareJoin (HashProbe.cpp:1639-1700,1826-1890):
        //   TRUE  = some build row with an equal key passes the filter;
        //   NULL  = none does, but the filter passes on some build row whose key is null (probe key
        //           not null: nullKeyProbeRows x listNullKeyRows) or on ANY build row (probe key
        //           null: crossJoinProbeRows x listAllRows);
        //   FALSE = neither. A null probe-side filter input makes the filter NULL for every pair
        //           (filterPropagateNulls): passes() is false for all of them, the row is FALSE.
        bool nullKey = false;
        for (auto& k : keys_) {
          nullKey = nullKey || k.isNull(cursorRow_);
        }
        char* first = nullptr;
        for (char* cur = nullKey ? nullptr : hit; cur && !first; cur = table_->table->nextRow(cur)) {
          if (passes(cursorRow_, cur)) {
            first = cur;
          }
        }
        bool isNull = false;
        if (!first) {
          for (auto* container : table_->containers) {
            for (char* row : container->rows()) {
              bool rowKeyNull = false;
              for (const auto& col : container->keys()) {
                rowKeyNull = rowKeyNull || row[col.nullOffset] != 0;
              }
              if ((nullKey || rowKeyNull) && passes(cursorRow_, row)) {
                isNull = true;
                break;
              }
            }
            if (isNull) {
              break;
            }
          }
        }
        if (joinType_ == VX355_JOIN_ANTI) {
          if (!first && !isNull) {
            emit(n++, cursorRow_, nullptr, mapping, buildRows, cols, colIds, numCols);
          }
        } else {
          emit(n++, cursorRow_, nullptr, mapping, buildRows, cols, colIds, numCols);
          if (buildRows) {
            buildRows[n - 1] = first ? static_cast<int32_t>(rowId(first)) : (isNull ? -2 : -1);
          }
        }
        ++cursorRow_;
        continue;
      }
      if (joinType_ == VX355_JOIN_ANTI && nullAware_) {
        // HashProbe.cpp:1316-1328 (no filter) + HashBuild's antiJoinHasNullKeys: a null on the
        // build side empties the result; an empty build side passes everything; else rows with
        // non-null keys and no match.
        bool out;
        if (table_->hasNullKeys) {
          out = false;
        } else if (table_->numRows == 0) {
          out = true;
        } else {
          out = !hit;
          for (auto& k : keys_) {
            if (k.isNull(cursorRow_)) {
              out = false;
            }
          }
        }
        if (out) {
          emit(n++, cursorRow_, nullptr, mapping, buildRows, cols, colIds, numCols);
        }
        ++cursorRow_;
        continue;
      }
      if (joinType_ == VX355_JOIN_LEFT_SEMI_PROJECT || joinType_ == VX355_JOIN_ANTI ||
          joinType_ == VX355_JOIN_LEFT_SEMI_FILTER) {
        // One output row at most: "does any pair of this probe row pass?" (rows with null keys
        // have no pair: regular anti returns them, core/PlanNode.h:3147-3150).
        char* first = nullptr;
        for (char* cur = hit; cur && !first; cur = table_->table->nextRow(cur)) {
          if (passes(cursorRow_, cur)) {
            first = cur;
          }
        }
        if (joinType_ == VX355_JOIN_LEFT_SEMI_PROJECT) {
          // every probe row once; the match column is "first != null" (not null aware)
          emit(n++, cursorRow_, nullptr, mapping, buildRows, cols, colIds, numCols);
          if (buildRows) {
            int32_t value = first ? static_cast<int32_t>(rowId(first)) : -1;
            if (nullAware_ && !first) {
              // fillLeftSemiProjectMatchColumn (HashProbe.cpp:923-966) without a filter; -2 = NULL:
              // empty build side: NULL if it held null keys, else FALSE; otherwise NULL for a null
              // probe key, and NULL instead of FALSE when the build side holds a null key
              bool nullKey = false;
              for (auto& k : keys_) {
                nullKey = nullKey || k.isNull(cursorRow_);
              }
              if (table_->numRows == 0) {
                value = table_->hasNullKeys ? -2 : -1;
              } else if (nullKey || table_->hasNullKeys) {
                value = -2;
              }
            }
            buildRows[n - 1] = value;
          }
        } else if ((first != nullptr) == (joinType_ == VX355_JOIN_LEFT_SEMI_FILTER)) {
          emit(n++, cursorRow_, nullptr, mapping, buildRows, cols, colIds, numCols);
        }
        ++cursorRow_;
        continue;
      }
      // inner / left / right / full: list the passing pairs of the chain
      char* cur = cursorChain_;
      if (!chainOpen_) {
        cur = hit;
        chainOpen_ = true;
        anyPass_ = false;
      }
      while (cur && n < maxRows) {
        if (passes(cursorRow_, cur)) {
          emit(n++, cursorRow_, cur, mapping, buildRows, cols, colIds, numCols);
          anyPass_ = true;
          if (marksProbed) {
            table_->probed[rowId(cur)] = 1;
          }
        }
        cur = table_->table->nextRow(cur);
      }
      if (cur) {
        cursorChain_ = cur;  // output page full in the middle of the chain
        break;
      }
      cursorChain_ = nullptr;
      if (!anyPass_ && includeMisses) {
        if (n >= maxRows) {
          break;  // the miss row opens the next page (NoMatchDetector's carried-over row)
        }
        emit(n++, cursorRow_, nullptr, mapping, buildRows, cols, colIds, numCols);
      }
      chainOpen_ = false;
      ++cursorRow_;
    }
    *nOut = n;
    *finished = cursorRow_ >= numRows_;
  }

  // HashProbe::getBuildSideOutput (HashProbe.cpp:993-1080), called by the last prober after
  // all probe input: right / full / right anti joins list the build rows no probe matched
  // (listNotProbedRows), right semi filter the matched ones (listProbedRows), right semi
  // project every row with its probed flag as the 'match' column (listAllRows +
  // extractProbedFlags; column id VX355_BUILD_COL_MATCH); container order = ascending row id.
  void getBuildSideOutput(int32_t maxRows, int32_t* buildRows, vx355_out_column* cols, const int32_t* colIds,
                          int32_t numCols, int32_t* nOut, int32_t* finished) {
    const bool wantProbed = joinType_ == VX355_JOIN_RIGHT_SEMI_FILTER;
    const bool all = joinType_ == VX355_JOIN_RIGHT_SEMI_PROJECT;
    int32_t n = 0;
    while (buildCursor_ < table_->numRows && n < maxRows) {
      if (all || (table_->probed[buildCursor_] != 0) == wantProbed) {
        char* row = table_->rowById(buildCursor_);
        if (buildRows) {
          buildRows[n] = static_cast<int32_t>(buildCursor_);
        }
        for (int32_t c = 0; c < numCols; ++c) {
          if (colIds[c] == VX355_BUILD_COL_MATCH) {
            const uint8_t match = table_->probed[buildCursor_] != 0;
            writeOut(cols[c], n, false, &match, 0);
          } else {
            extractStored(row, table_->containers[0]->deps()[colIds[c]], cols[c], n);
          }
        }
        ++n;
      }
      ++buildCursor_;
    }
    *nOut = n;
    *finished = buildCursor_ >= table_->numRows;
  }

 private:
  int64_t rowId(char* row) const {
    // Local allocation index, re-based at finish for merged containers.
    return *reinterpret_cast<int64_t*>(row + table_->containers[0]->rowIdOffset());
  }
  void emit(int32_t i, int32_t probeRow, char* buildRow, int32_t* mapping, int32_t* buildRows,
            vx355_out_column* cols, const int32_t* colIds, int32_t numCols) {
    mapping[i] = probeRow;
    if (buildRows) {
      buildRows[i] = buildRow ? static_cast<int32_t>(rowId(buildRow)) : -1;
    }
    for (int32_t c = 0; c < numCols; ++c) {
      const auto& col = table_->containers[0]->deps()[colIds[c]];
      if (!buildRow) {
        writeOut(cols[c], i, true, nullptr, kindWidth(col.kind));
      } else {
        extractStored(buildRow, col, cols[c], i);
      }
    }
  }

  struct Operand {
    bool null = false;
    int cls = 0;  // 0 int64, 1 double, 2 string
    int64_t i = 0;
    double d = 0;
    std::string s;
  };
  Operand probeOperand(int32_t col, int32_t row) const {
    Operand o;
    Decoded d(&batch_->cols[col]);
    if (d.isNull(row)) {
      o.null = true;
      return o;
    }
    const int32_t kind = batch_->cols[col].type_kind;
    if (kind == VX355_VARCHAR || kind == VX355_VARBINARY) {
      uint8_t tmp;
      auto* sv = static_cast<const StringView*>(d.valuePtr(row, &tmp));
      o.cls = 2;
      o.s.assign(sv->data(), sv->size);
    } else if (kind == VX355_REAL || kind == VX355_DOUBLE) {
      o.cls = 1;
      o.d = d.doubleAt(row);
    } else {
      o.i = d.int64At(row);
    }
    return o;
  }
  Operand buildOperand(int32_t dep, const char* row) const {
    Operand o;
    const auto& c = table_->containers[0]->deps()[dep];
    if (row[c.nullOffset] != 0) {
      o.null = true;
      return o;
    }
    const char* p = row + c.offset;
    if (c.kind == VX355_VARCHAR || c.kind == VX355_VARBINARY) {
      auto* sv = reinterpret_cast<const StringView*>(p);
      o.cls = 2;
      o.s.assign(sv->data(), sv->size);
    } else if (c.kind == VX355_REAL) {
      float f;
      std::memcpy(&f, p, 4);
      o.cls = 1;
      o.d = f;
    } else if (c.kind == VX355_DOUBLE) {
      o.cls = 1;
      std::memcpy(&o.d, p, 8);
    } else {
      std::memcpy(&o.i, p, 8);  // integer kinds are stored widened (extractStored)
    }
    return o;
  }
  // evalFilter (HashProbe.cpp:1713): true when every term holds for the pair; a null operand
  // makes the conjunct null, i.e. not passing.
  bool passes(int32_t probeRow, const char* buildRow) const {
    for (const auto& t : filter_) {
      Operand l = t.left_side == 0 ? probeOperand(t.left_col, probeRow) : buildOperand(t.left_col, buildRow);
      Operand r;
      if (t.right_kind == 1) {
        r = probeOperand(t.right_col, probeRow);
      } else if (t.right_kind == 2) {
        r = buildOperand(t.right_col, buildRow);
      } else if (t.const_kind == VX355_BIGINT) {
        r.i = t.i64;
      } else if (t.const_kind == VX355_DOUBLE) {
        r.cls = 1;
        r.d = t.f64;
      } else {
        r.cls = 2;
        r.s.assign(t.str, t.str_size);
      }
      if (l.null || r.null) {
        return false;
      }
      auto cmp = [&](auto a, auto b) { return compareLikeVelox(t.cmp, a, b); };
      bool ok;
      if (l.cls == 2 || r.cls == 2) {
        if (l.cls != r.cls || (t.cmp != VX355_CMP_EQ && t.cmp != VX355_CMP_NE)) {
          throw std::runtime_error("join filter: strings compare with strings, = and <> only");
        }
        ok = cmp(l.s, r.s);
      } else if (l.cls == 0 && r.cls == 0) {
        ok = cmp(l.i, r.i);
      } else {
        ok = cmp(l.cls == 0 ? static_cast<double>(l.i) : l.d, r.cls == 0 ? static_cast<double>(r.i) : r.d);
      }
      if (!ok) {
        return false;
      }
    }
    return true;
  }

  JoinTable* table_;
  int32_t joinType_;
  bool nullAware_;
  bool nullAsValue_ = false;
  std::vector<vx355_join_filter_term> filter_;
  const vx355_batch* batch_ = nullptr;  // the batch being probed (filter operands), borrowed
  bool chainOpen_ = false;
  bool anyPass_ = false;
  int64_t buildCursor_ = 0;
  std::vector<int32_t> keyCols_;
  std::vector<Decoded> keys_;
  HashLookup lookup_;
  int32_t numRows_ = 0;
  int32_t cursorRow_ = 0;
  char* cursorChain_ = nullptr;
};

}  // namespace orc

// ---------------------------------------------------------------------------
// C API
// ---------------------------------------------------------------------------

using namespace orc;

struct orc_hasher {
  VectorHasher h;
  explicit orc_hasher(int32_t k) : h(k) {}
};
struct orc_agg {
  Aggregation a;
  orc_agg(const vx355_agg_spec& s, bool ad) : a(s, ad) {}
};
struct orc_join_build {
  JoinBuild b;
  explicit orc_join_build(const vx355_join_build_spec& s) : b(s) {}
};
struct orc_join_table {
  JoinTable t;
};
struct orc_join_probe {
  JoinProbe p;
  orc_join_probe(JoinTable* t, const vx355_join_probe_spec& s) : p(t, s) {}
};

#define ORC_TRY try {
#define ORC_CATCH                          \
  }                                        \
  catch (const UserError& e) {             \
    gLastError = e.what();                 \
    return VX355_EUSER;                    \
  }                                        \
  catch (const std::exception& e) {        \
    gLastError = e.what();                 \
    return VX355_EINTERNAL;                \
  }                                        \
  return VX355_OK;

extern "C" {

uint64_t orc_twang_mix64(uint64_t key) { return twangMix64(key); }
uint32_t orc_twang_32from64(uint64_t key) { return twang32From64(key); }
uint32_t orc_jenkins_rev_mix32(uint32_t key) { return jenkinsRevMix32(key); }
uint64_t orc_hash_mix(uint64_t upper, uint64_t lower) { return hashMix(upper, lower); }
uint32_t orc_crc32c_u64(uint32_t checksum, uint64_t value) { return crc32U64(checksum, value); }
uint64_t orc_hash_bytes(uint64_t seed, const char* data, size_t size) {
  return hashBytes(seed, data, size);
}
uint32_t orc_xxh32_u32(uint32_t value, uint32_t seed) { return xxh32U32(value, seed); }
uint64_t orc_hash_value(int32_t type_kind, const void* value) { return hashValue(type_kind, value); }
const char* orc_last_error(void) { return gLastError.c_str(); }

void orc_set_sum_overflow_rule(int32_t rule) { gSumOverflowRule = rule ? 1 : 0; }

// ---- SplitBlockBloomFilter (common/base/SplitBlockBloomFilter.h) ----------------------
// A block is 'lanes' 32-bit words (one SIMD register of the host the reference
// runs on). makeSaltsVec (:96-121): 8 lanes use all salts, 4 lanes every other one.
static const uint32_t kBloomSalts[8] = {0x2df1424bU, 0x44974d91U, 0x47b6137bU, 0x5c6bfb31U,
                                        0x705495c7U, 0x8824ad5bU, 0x9efc4947U, 0xa2b7289dU};

int64_t orc_bloom_num_blocks(int64_t num_elements, double false_positive, int32_t lanes) {
  // SplitBlockBloomFilter.cpp:27-34.
  const int k = lanes;
  const int64_t numBits = static_cast<int64_t>(
      std::ceil(-k * num_elements / std::log(1 - std::pow(false_positive, 1.0 / k))));
  const int64_t blockBits = 8 * static_cast<int64_t>(sizeof(uint32_t)) * lanes;
  return (numBits + blockBits - 1) / blockBits;
}

static inline uint64_t bloomBlockIndex(uint64_t hash, int64_t numBlocks) {
  return ((hash >> 32) * static_cast<uint64_t>(numBlocks)) >> 32;  // :91-93
}

static inline uint32_t bloomLaneBit(uint64_t hash, int32_t lanes, int lane) {
  // makeMask(uint32_t hash) (:123-126): (salt * hash) >> 27 selects the bit of the lane.
  const uint32_t salt = lanes == 8 ? kBloomSalts[lane] : kBloomSalts[2 * lane];
  return 1u << ((salt * static_cast<uint32_t>(hash)) >> 27);
}

void orc_bloom_insert(uint32_t* blocks, int64_t num_blocks, int32_t lanes, const int64_t* values, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t h = twangMix64(static_cast<uint64_t>(values[i]));  // type/Filter.h:1354-1359
    uint32_t* block = blocks + bloomBlockIndex(h, num_blocks) * lanes;
    for (int l = 0; l < lanes; ++l) {
      block[l] |= bloomLaneBit(h, lanes, l);  // insert (:72-76)
    }
  }
}

void orc_bloom_test(const uint32_t* blocks, int64_t num_blocks, int32_t lanes, const int64_t* values, int64_t n,
                    uint8_t* may_contain_out) {
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t h = twangMix64(static_cast<uint64_t>(values[i]));
    const uint32_t* block = blocks + bloomBlockIndex(h, num_blocks) * lanes;
    bool all = true;
    for (int l = 0; l < lanes; ++l) {
      all = all && (block[l] & bloomLaneBit(h, lanes, l));  // mayContain (:81-88)
    }
    may_contain_out[i] = all ? 1 : 0;
  }
}

int orc_hash_columns(const vx355_batch* batch, const int32_t* key_cols, int32_t n_keys,
                     const uint64_t* rows, int32_t mix_first, uint64_t* out) {
  ORC_TRY
  for (int32_t i = 0; i < n_keys; ++i) {
    const vx355_column* col = &batch->cols[key_cols[i]];
    VectorHasher h(col->type_kind);
    Decoded d(col);
    h.hash(d, batch->num_rows, rows, i > 0 || mix_first, out);
  }
  ORC_CATCH
}

orc_hasher* orc_hasher_create(int32_t type_kind) { return new orc_hasher(type_kind); }
void orc_hasher_destroy(orc_hasher* h) { delete h; }
int orc_hasher_compute_value_ids(orc_hasher* h, const vx355_column* col, int32_t num_rows,
                                 const uint64_t* rows, uint64_t* result) {
  Decoded d(col);
  return h->h.computeValueIds(d, num_rows, rows, result) ? 1 : 0;
}
void orc_hasher_lookup_value_ids(const orc_hasher* h, const vx355_column* col, int32_t num_rows,
                                 uint64_t* rows_inout, uint64_t* result) {
  Decoded d(col);
  h->h.lookupValueIds(d, num_rows, rows_inout, result);
}
void orc_hasher_cardinality(orc_hasher* h, int32_t reserve_pct, uint64_t* as_range,
                            uint64_t* as_distinct) {
  h->h.cardinality(reserve_pct, *as_range, *as_distinct);
}
uint64_t orc_hasher_enable_value_range(orc_hasher* h, uint64_t multiplier, int32_t reserve_pct) {
  return h->h.enableValueRange(multiplier, reserve_pct);
}
uint64_t orc_hasher_enable_value_ids(orc_hasher* h, uint64_t multiplier, int32_t reserve_pct) {
  return h->h.enableValueIds(multiplier, reserve_pct);
}
void orc_hasher_merge(orc_hasher* h, const orc_hasher* other, uint64_t max_num_distinct) {
  h->h.merge(other->h, max_num_distinct);
}
void orc_hasher_get_state(const orc_hasher* h, orc_hasher_state* out) {
  out->is_range = h->h.isRange();
  out->has_range = h->h.hasRange();
  out->range_overflow = h->h.rangeOverflow();
  out->distinct_overflow = h->h.distinctOverflow();
  out->min = h->h.min();
  out->max = h->h.max();
  out->multiplier = h->h.multiplier();
  out->range_size = h->h.rangeSize();
  out->num_distinct = h->h.numDistinct();
}

// Range-mode value ids for several keys at once; the per-key arithmetic is
// VectorHasher.h:523-566 (tryMapToRange / valueId) and VectorHasher.cpp:196-224.
int orc_value_ids(const vx355_batch* batch, const int32_t* key_cols,
                  const vx355_value_id_spec* specs, int32_t n_keys, const uint64_t* rows,
                  int32_t lookup, uint64_t* result, uint64_t* rows_out, int32_t* all_mapped) {
  ORC_TRY
  const int32_t n = batch->num_rows;
  const int32_t words = (n + 63) / 64;
  std::vector<uint64_t> sel(words, ~0ULL);
  if (rows) {
    std::memcpy(sel.data(), rows, words * 8);
  }
  if (n & 63) {
    sel.back() &= (1ULL << (n & 63)) - 1;
  }
  bool mapped = true;
  for (int32_t k = 0; k < n_keys; ++k) {
    const vx355_column* col = &batch->cols[key_cols[k]];
    Decoded d(col);
    const auto& s = specs[k];
    const bool isString = isStringKind(col->type_kind);
    for (int32_t r = 0; r < n; ++r) {
      if (!bitSet(sel.data(), r)) {
        continue;
      }
      if (d.isNull(r)) {
        if (s.multiplier == 1) {
          result[r] = 0;
        }
        continue;
      }
      uint64_t id;
      if (col->type_kind == VX355_BOOLEAN) {
        id = d.int64At(r) ? 2 : 1;
      } else {
        int64_t v;
        bool ok = true;
        if (isString) {
          uint8_t tmp;
          auto* sv = static_cast<const StringView*>(d.valuePtr(r, &tmp));
          if (sv->size > VectorHasher::kStringASRangeMaxSize) {
            ok = false;
            v = 0;
          } else {
            v = VectorHasher::stringAsNumber(sv->data(), sv->size);
          }
        } else {
          v = d.int64At(r);
        }
        if (!ok || v < s.min || v > s.max) {
          if (lookup) {
            setBit(sel.data(), r, false);
          } else {
            mapped = false;
          }
          continue;
        }
        id = static_cast<uint64_t>(v) - static_cast<uint64_t>(s.min) + 1;
      }
      result[r] = s.multiplier == 1 ? id : result[r] + s.multiplier * id;
    }
  }
  if (lookup && rows_out) {
    std::memcpy(rows_out, sel.data(), words * 8);
  }
  if (all_mapped) {
    *all_mapped = mapped ? 1 : 0;
  }
  ORC_CATCH
}

// processFlatFilterResults (exec/OperatorUtils.cpp:231-257).
int orc_filter_compact(const uint64_t* values, const uint64_t* nulls, const uint64_t* rows,
                       int32_t num_rows, int32_t* idx_out, int32_t* n_out) {
  int32_t passed = 0;
  for (int32_t r = 0; r < num_rows; ++r) {
    if (bitSet(values, r) && (!nulls || bitSet(nulls, r)) && (!rows || bitSet(rows, r))) {
      idx_out[passed++] = r;
    }
  }
  *n_out = passed;
  return VX355_OK;
}

// HashPartitionFunction::partition (exec/HashPartitionFunction.cpp:76-118).
int orc_partition(const uint64_t* hashes, int32_t num_rows, int32_t kind, int32_t num_partitions,
                  int32_t bit_begin, int32_t bit_end, uint32_t* out) {
  const uint64_t mask = bit_end - bit_begin >= 64 ? ~0ULL : ((1ULL << (bit_end - bit_begin)) - 1);
  for (int32_t i = 0; i < num_rows; ++i) {
    uint64_t h = hashes[i];
    switch (kind) {
      case VX355_PART_MODULO:
        out[i] = static_cast<uint32_t>(h % static_cast<uint64_t>(num_partitions));
        break;
      case VX355_PART_BIT_RANGE:
        out[i] = static_cast<uint32_t>((h >> bit_begin) & mask);
        break;
      case VX355_PART_LOCAL_MODULO: {
        uint32_t l = xxh32U32(reverseBitsPerByte(static_cast<uint32_t>(h)), 0);
        out[i] = l % static_cast<uint32_t>(num_partitions);
        break;
      }
      case VX355_PART_LOCAL_BIT_RANGE: {
        uint32_t l = xxh32U32(reverseBitsPerByte(static_cast<uint32_t>(h)), 0);
        out[i] = static_cast<uint32_t>((static_cast<uint64_t>(l) >> bit_begin) & mask);
        break;
      }
      default:
        return VX355_EINVAL;
    }
  }
  return VX355_OK;
}

// PartitionedOutput's pages: see presto_page.h. Same contract as vx355_presto_serialize with
// host buffers.
int orc_presto_serialize(const vx355_batch* batch, const int32_t* rows, const int64_t* offsets, int32_t num_pages,
                         int32_t flags, void* out, int64_t out_capacity, int64_t* page_offsets) {
  ORC_TRY
  int64_t at = 0;
  for (int32_t p = 0; p < num_pages; ++p) {
    page_offsets[p] = at;
    if (offsets[p + 1] == offsets[p]) {
      continue;
    }
    std::vector<unsigned char> page;
    try {
      page = prestoPage(*batch, rows, offsets[p], offsets[p + 1], (flags & VX355_PAGE_CHECKSUM) != 0,
                        (flags & VX355_PAGE_LOSSLESS_TIMESTAMP) != 0);
    } catch (const UserErrorTag& e) {
      throw UserError(e.what());
    }
    if (out) {
      if (at + static_cast<int64_t>(page.size()) > out_capacity) {
        throw std::runtime_error("orc_presto_serialize: output buffer too small");
      }
      std::memcpy(static_cast<char*>(out) + at, page.data(), page.size());
    }
    at += static_cast<int64_t>(page.size());
  }
  page_offsets[num_pages] = at;
  ORC_CATCH
}

// FilterProject::filter + project (exec/FilterProject.cpp:102-275) for
// conjunctions of column-vs-constant comparisons and products of affine
// factors; a null input fails a comparison (exec/OperatorUtils.cpp:240-248) and
// nulls a projection.
int orc_filter_project(const vx355_batch* batch, const vx355_filter_term* terms, int32_t n_terms,
                       const vx355_projection* projections, int32_t n_projections, int32_t* idx_out,
                       int32_t* n_out, double* const* proj_out, uint64_t* const* proj_nulls_out) {
  ORC_TRY
  const int32_t n = batch->num_rows;
  int32_t passed = 0;
  for (int32_t r = 0; r < n; ++r) {
    bool ok = true;
    for (int32_t t = 0; t < n_terms && ok; ++t) {
      const auto& term = terms[t];
      Decoded d(&batch->cols[term.col]);
      if (d.isNull(r)) {
        ok = false;
        break;
      }
      auto cmp = [&](auto a, auto b) { return compareLikeVelox(term.cmp, a, b); };
      if (term.const_kind == VX355_BIGINT) {
        ok = cmp(d.int64At(r), term.i64);
      } else if (term.const_kind == VX355_DOUBLE) {
        ok = cmp(d.doubleAt(r), term.f64);
      } else {
        uint8_t tmp;
        auto* sv = static_cast<const StringView*>(d.valuePtr(r, &tmp));
        bool eq = sv->size == static_cast<uint32_t>(term.str_size) &&
            std::memcmp(sv->data(), term.str, sv->size) == 0;
        ok = term.cmp == VX355_CMP_EQ ? eq : !eq;
      }
    }
    if (!ok) {
      continue;
    }
    for (int32_t j = 0; j < n_projections; ++j) {
      const auto& p = projections[j];
      bool valid = true;
      double acc = 0;
      for (int32_t f = 0; f < p.num_factors; ++f) {
        double v = p.factors[f].offset;
        if (p.factors[f].col >= 0) {
          Decoded d(&batch->cols[p.factors[f].col]);
          if (d.isNull(r)) {
            valid = false;
            continue;
          }
          v = p.factors[f].scale * d.doubleAt(r) + p.factors[f].offset;
        }
        acc = f == 0 ? v : acc * v;
      }
      proj_out[j][passed] = valid ? acc : 0.0;
      if (proj_nulls_out && proj_nulls_out[j]) {
        setBit(proj_nulls_out[j], passed, valid);
      }
    }
    idx_out[passed++] = r;
  }
  *n_out = passed;
  ORC_CATCH
}

int orc_agg_create(const vx355_agg_spec* spec, int32_t hash_adaptivity, orc_agg** out) {
  ORC_TRY
  *out = new orc_agg(*spec, hash_adaptivity != 0);
  ORC_CATCH
}
int orc_agg_add_input(orc_agg* h, const vx355_batch* batch) {
  ORC_TRY
  h->a.addInput(*batch);
  ORC_CATCH
}
int orc_agg_no_more_input(orc_agg* h) {
  h->a.noMoreInput();
  return VX355_OK;
}
int orc_agg_get_output(orc_agg* h, vx355_out_column* cols, int32_t num_cols, int32_t max_rows,
                       int32_t* n_out, int32_t* finished) {
  ORC_TRY
  h->a.getOutput(cols, num_cols, max_rows, n_out, finished);
  ORC_CATCH
}
int orc_agg_get_stats(const orc_agg* h, vx355_agg_stats* out) {
  h->a.stats(out);
  return VX355_OK;
}
void orc_agg_destroy(orc_agg* h) { delete h; }

int orc_join_build_create(const vx355_join_build_spec* spec, orc_join_build** out) {
  ORC_TRY
  *out = new orc_join_build(*spec);
  ORC_CATCH
}
int orc_join_build_add_input(orc_join_build* h, const vx355_batch* batch) {
  ORC_TRY
  h->b.addInput(*batch);
  ORC_CATCH
}
// HashBuild::finishHashBuild (HashBuild.cpp:819-993): the last driver steals
// its peers' tables and builds one table.
int orc_join_build_finish(orc_join_build* h, orc_join_build* const* others, int32_t num_others,
                          orc_join_table** out) {
  ORC_TRY
  auto* t = new orc_join_table();
  t->t.table = std::move(h->b.table_);
  t->t.depKinds = h->b.depKinds_;
  std::vector<HashTable*> raw;
  int64_t base = 0;
  t->t.containers.push_back(t->t.table->rows());
  t->t.containerBase.push_back(0);
  base += t->t.table->rows()->numRows();
  for (int32_t i = 0; i < num_others; ++i) {
    t->t.others.push_back(std::move(others[i]->b.table_));
    auto* other = t->t.others.back().get();
    raw.push_back(other);
    // Re-number the merged rows after this table's rows.
    for (char* row : other->rows()->rows()) {
      *reinterpret_cast<int64_t*>(row + other->rows()->rowIdOffset()) += base;
    }
    t->t.containers.push_back(other->rows());
    t->t.containerBase.push_back(base);
    base += other->rows()->numRows();
  }
  t->t.numRows = base;
  t->t.probed.assign(static_cast<size_t>(base), 0);
  t->t.hasNullKeys = h->b.hasNullKeys_;
  for (int32_t i = 0; i < num_others; ++i) {
    t->t.hasNullKeys = t->t.hasNullKeys || others[i]->b.hasNullKeys_;
  }
  t->t.table->setBuildThreads(gJoinBuildThreads);
  t->t.table->prepareJoinTable(raw);
  *out = t;
  ORC_CATCH
}
// Threads of HashTable::parallelJoinBuild for the join tables finished after this call (1 = the
// serial insert, the default: what the parity tests run). bench.py's multi-thread CPU leg only.
void orc_set_join_build_threads(int32_t n) { gJoinBuildThreads = n < 1 ? 1 : n; }
void orc_join_build_destroy(orc_join_build* h) { delete h; }
void orc_join_table_release(orc_join_table* t) { delete t; }
int orc_join_table_get_stats(const orc_join_table* t, vx355_join_table_stats* out) {
  out->num_rows = t->t.numRows;
  out->num_distinct = t->t.table->numDistinctKeys();
  out->capacity = static_cast<int64_t>(t->t.table->capacity());
  out->hash_mode = static_cast<int32_t>(t->t.table->hashMode());
  out->has_duplicates = t->t.table->hasDuplicates();
  return VX355_OK;
}
int orc_join_probe_create(orc_join_table* t, const vx355_join_probe_spec* spec,
                          orc_join_probe** out) {
  ORC_TRY
  switch (spec->join_type) {
    case VX355_JOIN_INNER:
    case VX355_JOIN_LEFT:
    case VX355_JOIN_LEFT_SEMI_FILTER:
    case VX355_JOIN_ANTI:
    case VX355_JOIN_RIGHT:
    case VX355_JOIN_FULL:
    case VX355_JOIN_RIGHT_SEMI_FILTER:
    case VX355_JOIN_LEFT_SEMI_PROJECT:
    case VX355_JOIN_COUNTING_LEFT_SEMI_FILTER:
    case VX355_JOIN_COUNTING_ANTI:
    case VX355_JOIN_RIGHT_SEMI_PROJECT:
    case VX355_JOIN_RIGHT_ANTI:
      break;
    default:
      gLastError = "join type not restated in the oracle";
      return VX355_EUNSUPPORTED;
  }
  if (spec->null_aware && spec->join_type != VX355_JOIN_ANTI && spec->join_type != VX355_JOIN_LEFT_SEMI_PROJECT) {
    gLastError = "null-aware semantics restated for the anti and the left semi project join";
    return VX355_EUNSUPPORTED;
  }
  *out = new orc_join_probe(&t->t, *spec);
  ORC_CATCH
}
int orc_join_probe_set_filter(orc_join_probe* h, const vx355_join_filter_term* terms, int32_t n_terms) {
  ORC_TRY
  h->p.setFilter(terms, n_terms);
  ORC_CATCH
}
int orc_join_probe_add_input(orc_join_probe* h, const vx355_batch* batch) {
  ORC_TRY
  h->p.addInput(*batch);
  ORC_CATCH
}
int orc_join_probe_get_output(orc_join_probe* h, int32_t max_rows, int32_t* mapping_out,
                              int32_t* build_rows_out, vx355_out_column* build_cols,
                              const int32_t* build_col_ids, int32_t num_build_cols,
                              int32_t* n_out, int32_t* finished) {
  ORC_TRY
  h->p.getOutput(max_rows, mapping_out, build_rows_out, build_cols, build_col_ids, num_build_cols,
                 n_out, finished);
  ORC_CATCH
}
int orc_join_probe_get_build_side_output(orc_join_probe* h, int32_t max_rows, int32_t* build_rows_out,
                                         vx355_out_column* build_cols, const int32_t* build_col_ids,
                                         int32_t num_build_cols, int32_t* n_out, int32_t* finished) {
  ORC_TRY
  h->p.getBuildSideOutput(max_rows, build_rows_out, build_cols, build_col_ids, num_build_cols, n_out, finished);
  ORC_CATCH
}
void orc_join_probe_destroy(orc_join_probe* h) { delete h; }

}  // extern "C"
