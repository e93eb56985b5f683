// vx355_filter_project: FilterProject (exec/FilterProject.cpp:102-275) for the
// TPC-H Q1 / Q3 expression class, HBM resident. Three launches: filter bits
// (one ballot per 64 rows), order-preserving compaction to selected row
// numbers (processFilterResults, exec/OperatorUtils.cpp:231-257), projections
// over the selected rows.
#include "common.h"
#include "expr_device.h"

namespace vx {

void compactBits(const uint64_t* dValues, const uint64_t* dNulls, const uint64_t* dRows,
                 int64_t numRows, int32_t* dOut, DevBuf& scratch, int64_t* total);

namespace {

struct FilterArgs {
  TermArg terms[kMaxTerms];
  int32_t numTerms;
  int64_t numRows;
  uint64_t* bits;
};

__global__ __launch_bounds__(256) void k_filter_bits(FilterArgs a) {
  const int64_t numWords = (a.numRows + 63) >> 6;
  const int64_t waveStride = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t w = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; w < numWords;
       w += waveStride) {
    const int64_t row = (w << 6) + lane();
    const bool pass = row < a.numRows && evalFilter(a.terms, a.numTerms, row);
    const uint64_t m = ballot(pass);
    if (lane() == 0) {
      a.bits[w] = m;
    }
  }
}

// One comparison of a flat, null-free fixed-width column against a constant:
// the filters of TPC-H Q1 / Q3 (l_shipdate, o_orderdate). No per-row plan
// interpretation: one load, one compare, one ballot per 64 rows.
template <typename T>
__global__ __launch_bounds__(256) void k_filter_bits_flat(const T* values, T constant, int32_t cmp,
                                                           int64_t numRows, uint64_t* bits) {
  // A wave covers 4 consecutive selection words per step: four independent
  // coalesced loads in flight per lane.
  constexpr int kWords = 4;
  const int64_t numWords = (numRows + 63) >> 6;
  const int64_t numGroups = (numWords + kWords - 1) / kWords;
  const int64_t waveStride = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t g = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; g < numGroups;
       g += waveStride) {
    T v[kWords];
#pragma unroll
    for (int j = 0; j < kWords; ++j) {
      const int64_t row = ((g * kWords + j) << 6) + lane();
      v[j] = row < numRows ? values[row] : T();
    }
#pragma unroll
    for (int j = 0; j < kWords; ++j) {
      const int64_t w = g * kWords + j;
      const int64_t row = (w << 6) + lane();
      const uint64_t m = ballot(row < numRows && compareValues<T>(cmp, v[j], constant));
      if (lane() == 0 && w < numWords) {
        bits[w] = m;
      }
    }
  }
}

struct ProjectArgs {
  ProjectionArg proj[kMaxProjections];
  double* out[kMaxProjections];
  uint64_t* outNulls[kMaxProjections];
  int32_t numProj;
  const int32_t* rows;
  int64_t count;
};

__global__ __launch_bounds__(256) void k_project(ProjectArgs a) {
  const int64_t numWords = (a.count + 63) >> 6;
  const int64_t waveStride = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t w = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; w < numWords;
       w += waveStride) {
    const int64_t p = (w << 6) + lane();
    const bool active = p < a.count;
    const int64_t row = active ? a.rows[p] : 0;
    for (int j = 0; j < a.numProj; ++j) {
      bool valid = active;
      double v = 0;
      if (active) {
        v = evalProjection(a.proj[j], row, &valid);
        a.out[j][p] = valid ? v : 0.0;
      }
      if (a.outNulls[j]) {
        const uint64_t m = ballot(valid);
        if (lane() == 0) {
          a.outNulls[j][w] = m;
        }
      }
    }
  }
}

}  // namespace

// Host-side translation of the ABI structs into kernel arguments; also used by
// the fused aggregation input.
void makeTermArgs(const DeviceBatch& db, const vx355_filter_term* terms, int32_t n, TermArg* out) {
  VX_CHECK_ARG(n >= 0 && n <= kMaxTerms, "at most 4 filter terms");
  for (int32_t i = 0; i < n; ++i) {
    const auto& t = terms[i];
    VX_CHECK_ARG(t.cmp >= VX355_CMP_EQ && t.cmp <= VX355_CMP_GE, "bad comparison");
    TermArg& a = out[i];
    a = TermArg{};
    a.col = db.col(t.col);
    a.cmp = t.cmp;
    a.constKind = t.const_kind;
    const int32_t kind = a.col.kind;
    if (t.const_kind == VX355_BIGINT) {
      if (!isIntLike(kind)) {
        VX_THROW(VX355_EUNSUPPORTED, "integer constant against a non-integer column");
      }
      a.i64 = t.i64;
    } else if (t.const_kind == VX355_DOUBLE) {
      if (kind != VX355_REAL && kind != VX355_DOUBLE) {
        VX_THROW(VX355_EUNSUPPORTED, "double constant against a non-floating column");
      }
      a.f64 = t.f64;
    } else if (t.const_kind == VX355_VARCHAR) {
      if (!isString(kind) || (t.cmp != VX355_CMP_EQ && t.cmp != VX355_CMP_NE) || t.str_size < 0 ||
          t.str_size > 12) {
        VX_THROW(VX355_EUNSUPPORTED, "string filter: only = / <> against a constant of <= 12 bytes");
      }
      unsigned char buf[12] = {0};
      std::memcpy(buf, t.str, t.str_size);
      a.strSize = static_cast<uint32_t>(t.str_size);
      std::memcpy(&a.strPrefix, buf, 4);
      std::memcpy(&a.strTail, buf + 4, 8);
    } else {
      VX_THROW(VX355_EINVAL, "bad const_kind");
    }
  }
}

void makeProjectionArgs(const DeviceBatch& db, const vx355_projection* proj, int32_t n,
                        ProjectionArg* out) {
  VX_CHECK_ARG(n >= 0 && n <= kMaxProjections, "at most 4 projections");
  for (int32_t j = 0; j < n; ++j) {
    const auto& p = proj[j];
    VX_CHECK_ARG(p.num_factors >= 1 && p.num_factors <= kMaxFactors, "1..4 factors per projection");
    ProjectionArg& a = out[j];
    a = ProjectionArg{};
    a.numFactors = p.num_factors;
    for (int f = 0; f < p.num_factors; ++f) {
      FactorArg& fa = a.factors[f];
      fa.scale = p.factors[f].scale;
      fa.offset = p.factors[f].offset;
      if (p.factors[f].col >= 0) {
        fa.hasCol = 1;
        fa.col = db.col(p.factors[f].col);
        const int32_t kind = fa.col.kind;
        if (!(isIntLike(kind) || kind == VX355_REAL || kind == VX355_DOUBLE)) {
          VX_THROW(VX355_EUNSUPPORTED, "projection over a non-numeric column");
        }
      }
    }
  }
}

}  // namespace vx

using namespace vx;

extern "C" int vx355_filter_project(const vx355_batch* batch, const vx355_filter_term* terms,
                                    int32_t n_terms, const vx355_projection* projections,
                                    int32_t n_projections, int32_t* idx_out, int32_t* n_out,
                                    double* const* proj_out, uint64_t* const* proj_nulls_out,
                                    int32_t out_mem) {
  VX_API_BEGIN
  auto& rt = Runtime::get();
  rt.requireInit();
  VX_CHECK_ARG(batch && n_out, "NULL argument");
  VX_CHECK_ARG(n_terms == 0 || terms, "terms is NULL");
  VX_CHECK_ARG(n_projections == 0 || (projections && proj_out), "projections / proj_out is NULL");
  *n_out = 0;
  std::vector<int32_t> used;
  for (int32_t i = 0; i < n_terms; ++i) {
    used.push_back(terms[i].col);
  }
  for (int32_t j = 0; j < n_projections; ++j) {
    VX_CHECK_ARG(projections[j].num_factors >= 1 && projections[j].num_factors <= kMaxFactors,
                 "1..4 factors per projection");
    for (int f = 0; f < projections[j].num_factors; ++f) {
      used.push_back(projections[j].factors[f].col);
    }
  }
  DeviceBatch db;
  db.load(batch, used);
  const int64_t n = db.numRows();
  if (n == 0) {
    return VX355_OK;
  }
  VX_CHECK_ARG(idx_out != nullptr, "idx_out is NULL");
  const bool host = out_mem == VX355_MEM_HOST;
  DevBuf bitsBuf, idxBuf, scratch, projBuf;
  const int64_t words = ceilDiv(n, 64);
  FilterArgs fa{};
  makeTermArgs(db, terms, n_terms, fa.terms);
  fa.numTerms = n_terms;
  fa.numRows = n;
  fa.bits = static_cast<uint64_t*>(bitsBuf.ensure(static_cast<size_t>(words) * 8 + 64));
  const TermArg* t0 = n_terms == 1 ? &fa.terms[0] : nullptr;
  const bool flat = t0 && t0->col.enc == VX355_FLAT && t0->col.nulls == nullptr;
  const int fgrid = streamGrid(words * 64, 256);
  if (flat && t0->constKind == VX355_BIGINT && t0->col.kind == VX355_INTEGER && t0->i64 >= INT32_MIN &&
      t0->i64 <= INT32_MAX) {
    VX_LAUNCH("k_filter_bits", k_filter_bits_flat<int32_t>, fgrid, 256, 0,
              static_cast<const int32_t*>(t0->col.values), static_cast<int32_t>(t0->i64), t0->cmp, n, fa.bits);
  } else if (flat && t0->constKind == VX355_BIGINT && t0->col.kind == VX355_BIGINT) {
    VX_LAUNCH("k_filter_bits", k_filter_bits_flat<int64_t>, fgrid, 256, 0,
              static_cast<const int64_t*>(t0->col.values), t0->i64, t0->cmp, n, fa.bits);
  } else if (flat && t0->constKind == VX355_DOUBLE && t0->col.kind == VX355_DOUBLE) {
    VX_LAUNCH("k_filter_bits", k_filter_bits_flat<double>, fgrid, 256, 0,
              static_cast<const double*>(t0->col.values), t0->f64, t0->cmp, n, fa.bits);
  } else {
    VX_LAUNCH("k_filter_bits", k_filter_bits, fgrid, 256, 0, fa);
  }
  int32_t* dIdx = host ? static_cast<int32_t*>(idxBuf.ensure(static_cast<size_t>(n) * 4 + 64)) : idx_out;
  int64_t passed = 0;
  compactBits(fa.bits, nullptr, nullptr, n, dIdx, scratch, &passed);
  if (n_projections > 0 && passed > 0) {
    ProjectArgs pa{};
    makeProjectionArgs(db, projections, n_projections, pa.proj);
    pa.numProj = n_projections;
    pa.rows = dIdx;
    pa.count = passed;
    const size_t valBytes = (static_cast<size_t>(passed) * 8 + 63) & ~static_cast<size_t>(63);
    const size_t nullBytes = (static_cast<size_t>(ceilDiv(passed, 64)) * 8 + 63) & ~static_cast<size_t>(63);
    char* base = nullptr;
    if (host) {
      base = static_cast<char*>(projBuf.ensure((valBytes + nullBytes) * n_projections + 64));
    }
    for (int32_t j = 0; j < n_projections; ++j) {
      VX_CHECK_ARG(proj_out[j] != nullptr, "proj_out[j] is NULL");
      uint64_t* nullsOut = proj_nulls_out ? proj_nulls_out[j] : nullptr;
      if (host) {
        pa.out[j] = reinterpret_cast<double*>(base + (valBytes + nullBytes) * j);
        pa.outNulls[j] = nullsOut ? reinterpret_cast<uint64_t*>(base + (valBytes + nullBytes) * j + valBytes)
                                  : nullptr;
      } else {
        pa.out[j] = proj_out[j];
        pa.outNulls[j] = nullsOut;
      }
    }
    VX_LAUNCH("k_project", k_project, streamGrid(ceilDiv(passed, 64) * 64, 256), 256, 0, pa);
    if (host) {
      for (int32_t j = 0; j < n_projections; ++j) {
        copyOut(proj_out[j], VX355_MEM_HOST, pa.out[j], static_cast<size_t>(passed) * 8);
        if (pa.outNulls[j]) {
          copyOut(proj_nulls_out[j], VX355_MEM_HOST, pa.outNulls[j],
                  static_cast<size_t>(ceilDiv(passed, 64)) * 8);
        }
      }
    }
  }
  if (host) {
    copyOut(idx_out, VX355_MEM_HOST, dIdx, static_cast<size_t>(passed) * 4);
  }
  rt.sync();
  *n_out = static_cast<int32_t>(passed);
  VX_API_END
}
