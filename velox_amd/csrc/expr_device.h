// Device evaluator for the small expression class of vx355_filter_project
// (include/vx355.h): conjunctions of column-vs-constant comparisons and
// products of affine factors in DOUBLE. Shared by the standalone FilterProject
// kernels and by HashAggregation's fused input path.
#pragma once
#include "device_utils.h"

namespace vx {

constexpr int kMaxTerms = 4;
constexpr int kMaxProjections = 4;
constexpr int kMaxFactors = 4;

struct TermArg {
  ColView col;
  int32_t cmp;
  int32_t constKind;  // VX355_BIGINT / VX355_DOUBLE / VX355_VARCHAR
  int64_t i64;
  double f64;
  uint32_t strSize;
  uint32_t strPrefix;
  uint64_t strTail;
};

struct FactorArg {
  ColView col;
  int32_t hasCol;
  int32_t pad;
  double scale;
  double offset;
};

struct ProjectionArg {
  FactorArg factors[kMaxFactors];
  int32_t numFactors;
  int32_t pad;
};

// The reference's comparison functions (functions/prestosql/Comparisons.h:42-121 ->
// util::floating_point::NaNAware*): for DOUBLE, NaN equals NaN and is greater than everything else.
template <typename T>
__device__ inline bool compareValues(int32_t cmp, T a, T b) {
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
template <>
__device__ inline bool compareValues<double>(int32_t cmp, double a, double b) {
  const bool an = a != a, bn = b != b;
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
}

// One filter term on one row; a null input fails the filter.
__device__ inline bool evalTerm(const TermArg& t, int64_t row) {
  if (colIsNull(t.col, row)) {
    return false;
  }
  const int64_t i = colIndex(t.col, row);
  if (t.constKind == VX355_BIGINT) {
    return compareValues<int64_t>(t.cmp, loadInt64(t.col, i), t.i64);
  }
  if (t.constKind == VX355_DOUBLE) {
    return compareValues<double>(t.cmp, loadDouble(t.col, i), t.f64);
  }
  // Inline string equality: size, 4-byte prefix and 8-byte tail all match
  // (unused bytes of an inline StringView are zero, type/StringView.h:76-98).
  const StringView16 v = loadView(t.col, i);
  const bool eq = v.size == t.strSize && v.prefix == t.strPrefix &&
      (v.size <= 4 || v.tail == t.strTail);
  return t.cmp == VX355_CMP_EQ ? eq : !eq;
}

__device__ inline bool evalFilter(const TermArg* terms, int numTerms, int64_t row) {
  for (int k = 0; k < numTerms; ++k) {
    if (!evalTerm(terms[k], row)) {
      return false;
    }
  }
  return true;
}

// f0 * f1 * ..., left to right; *valid is cleared when any input is null.
__device__ inline double evalProjection(const ProjectionArg& p, int64_t row, bool* valid) {
  double acc = 0;
  for (int f = 0; f < p.numFactors; ++f) {
    const FactorArg& fa = p.factors[f];
    double v = fa.offset;
    if (fa.hasCol) {
      if (colIsNull(fa.col, row)) {
        *valid = false;
        continue;
      }
      const double x = loadDouble(fa.col, colIndex(fa.col, row));
      // scale is +1 or -1 in TPC-H; keep the generic form exact for those:
      // 1 * x == x and -1 * x == -x bit for bit.
      v = fa.scale * x + fa.offset;
    }
    acc = f == 0 ? v : acc * v;
  }
  return acc;
}

}  // namespace vx
