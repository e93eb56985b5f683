// C++ consumer of the drop-in boundary, written the way the reference's operator tests are
// (exec/tests/AggregationTest.cpp, HashJoinTest.cpp: feed RowVectors, drain the operator,
// compare with an expected result computed independently). Expected values come from plain
// loops over the same host data — this program never touches the oracle.
//   g++ -std=c++17 -I include tests/cpp/operator_test.cpp -L velox_amd -lvx355 -Wl,-rpath,$PWD/velox_amd
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <random>
#include <set>
#include <thread>
#include <array>
#include <atomic>
#include <vector>

#include "vx355.hpp"

#define EXPECT(cond)                                                      \
  do {                                                                    \
    if (!(cond)) {                                                        \
      std::fprintf(stderr, "%s:%d: EXPECT(%s) failed\n", __FILE__, __LINE__, #cond); \
      return 1;                                                           \
    }                                                                     \
  } while (0)

namespace {

vx355_column flat(int32_t kind, const void* values, const uint64_t* nulls = nullptr) {
  vx355_column c{};
  c.type_kind = kind;
  c.encoding = VX355_FLAT;
  c.values = values;
  c.nulls = nulls;
  c.mem = VX355_MEM_HOST;
  return c;
}

struct OutCol {
  std::vector<char> values;
  std::vector<uint64_t> nulls;
  vx355_out_column desc{};
  OutCol(int32_t kind, int32_t width, int32_t capacity) : values(size_t(capacity) * width), nulls(capacity / 64 + 1) {
    desc.type_kind = kind;
    desc.mem = VX355_MEM_HOST;
    desc.values = values.data();
    desc.nulls = nulls.data();
  }
  template <typename T>
  T at(int32_t i) const {
    T v;
    std::memcpy(&v, values.data() + size_t(i) * sizeof(T), sizeof(T));
    return v;
  }
  bool valid(int32_t i) const { return (nulls[i >> 6] >> (i & 63)) & 1; }
};

int testAggregation() {
  // SELECT k, sum(v), count(*), min(v), max(v), avg(v) GROUP BY k — batches of 10 000 rows like
  // exec/benchmarks/SimpleAggregates.cpp:33-34; v = multiples of 1/8 so DOUBLE sums are exact.
  std::mt19937_64 rng(7);
  const int kBatches = 30, kRows = 10000;
  vx355::HashAggregation op({0}, {VX355_BIGINT},
                            {{VX355_AGG_SUM, 1, -1, VX355_DOUBLE, -1},
                             {VX355_AGG_COUNT_STAR, -1, -1, VX355_BIGINT, -1},
                             {VX355_AGG_MIN, 1, -1, VX355_DOUBLE, -1},
                             {VX355_AGG_MAX, 1, -1, VX355_DOUBLE, -1},
                             {VX355_AGG_AVG, 1, -1, VX355_DOUBLE, -1}});
  struct Acc {
    double sum = 0, mn = INFINITY, mx = -INFINITY;
    int64_t count = 0, seen = 0;
  };
  std::map<int64_t, Acc> expected;
  std::vector<int64_t> firstSeen;
  for (int b = 0; b < kBatches; ++b) {
    std::vector<int64_t> k(kRows);
    std::vector<double> v(kRows);
    std::vector<uint64_t> vNulls((kRows + 63) / 64, ~0ULL);
    for (int i = 0; i < kRows; ++i) {
      k[i] = int64_t(rng() % 1000) - 300;
      v[i] = double(rng() % 4096) / 8.0;
      if (rng() % 10 == 0) {
        vNulls[i >> 6] &= ~(1ULL << (i & 63));
      }
      auto it = expected.find(k[i]);
      if (it == expected.end()) {
        firstSeen.push_back(k[i]);
        it = expected.emplace(k[i], Acc{}).first;
      }
      ++it->second.count;
      if ((vNulls[i >> 6] >> (i & 63)) & 1) {
        it->second.sum += v[i];
        it->second.mn = std::min(it->second.mn, v[i]);
        it->second.mx = std::max(it->second.mx, v[i]);
        ++it->second.seen;
      }
    }
    vx355_column cols[2] = {flat(VX355_BIGINT, k.data()), flat(VX355_DOUBLE, v.data(), vNulls.data())};
    EXPECT(op.needsInput());
    op.addInput(vx355_batch{kRows, 2, cols});
  }
  op.noMoreInput();
  EXPECT(!op.needsInput());
  EXPECT((op.outputTypes() == std::vector<int32_t>{VX355_BIGINT, VX355_DOUBLE, VX355_BIGINT, VX355_DOUBLE,
                                                   VX355_DOUBLE, VX355_DOUBLE}));
  size_t row = 0;
  const int32_t kMax = 333;
  while (!op.isFinished()) {
    OutCol key(VX355_BIGINT, 8, kMax), sum(VX355_DOUBLE, 8, kMax), cnt(VX355_BIGINT, 8, kMax), mn(VX355_DOUBLE, 8, kMax),
        mx(VX355_DOUBLE, 8, kMax), avg(VX355_DOUBLE, 8, kMax);
    vx355_out_column out[6] = {key.desc, sum.desc, cnt.desc, mn.desc, mx.desc, avg.desc};
    const int32_t n = op.getOutput(out, 6, kMax);
    EXPECT(n <= kMax);
    for (int32_t i = 0; i < n; ++i, ++row) {
      EXPECT(row < firstSeen.size() && key.at<int64_t>(i) == firstSeen[row]);  // groups in first-seen order
      const Acc& e = expected.at(firstSeen[row]);
      EXPECT(cnt.at<int64_t>(i) == e.count);
      EXPECT(sum.valid(i) == (e.seen > 0));
      if (e.seen > 0) {
        EXPECT(sum.at<double>(i) == e.sum && mn.at<double>(i) == e.mn && mx.at<double>(i) == e.mx);
        EXPECT(avg.at<double>(i) == e.sum / double(e.seen));
      }
    }
  }
  EXPECT(row == firstSeen.size());
  EXPECT(op.stats().num_groups == int64_t(firstSeen.size()) && op.stats().input_rows == int64_t(kBatches) * kRows);
  // sum(BIGINT) overflow is a user error (SumAggregate.cpp:24), as in AggregationTest's overflow cases.
  vx355::HashAggregation ovf({}, {}, {{VX355_AGG_SUM, 0, -1, VX355_BIGINT, -1}});
  std::vector<int64_t> big(4, INT64_MAX / 2);
  vx355_column c = flat(VX355_BIGINT, big.data());
  bool threw = false;
  try {
    ovf.addInput(vx355_batch{4, 1, &c});
    ovf.noMoreInput();
    // the exact (128-bit) total is checked when it is read out: the one rule that does not depend
    // on the order the GPU adds in
    int64_t sum = 0;
    uint64_t valid = 0;
    vx355_out_column oc{VX355_BIGINT, VX355_MEM_HOST, &sum, &valid};
    ovf.getOutput(&oc, 1, 16);
  } catch (const vx355::UserError& e) {
    threw = std::string(e.what()).find("overflow") != std::string::npos;
  }
  EXPECT(threw);
  return 0;
}

int testSemiJoinWithoutDuplicates() {
  // HashBuild with dropDuplicates (HashBuild.cpp:517-548): a left semi join needs one build row per key; the
  // table then has no duplicate chains, the probe lists every probe row with a match once. Plus the
  // composition of two index vectors (wrapChild over a wrapped vector) on host memory.
  std::mt19937_64 rng(5);
  const int kBuild = 5000, kProbe = 7000;
  std::vector<int64_t> bk(kBuild), pk(kProbe);
  std::set<int64_t> keys;
  for (auto& k : bk) {
    k = rng() % 900;
    keys.insert(k);
  }
  for (auto& k : pk) {
    k = rng() % 1800;
  }
  vx355::HashBuild build({0}, {VX355_BIGINT}, {}, {}, VX355_JOIN_LEFT_SEMI_FILTER, false, false, /*dropDuplicates=*/true);
  vx355_column bc = flat(VX355_BIGINT, bk.data());
  build.addInput(vx355_batch{kBuild, 1, &bc});
  vx355::JoinTable table = build.noMoreInput();
  EXPECT(table.stats().has_duplicates == 0);
  vx355::HashProbe probe(table, {0}, VX355_JOIN_LEFT_SEMI_FILTER);
  vx355_column pc = flat(VX355_BIGINT, pk.data());
  probe.addInput(vx355_batch{kProbe, 1, &pc});
  std::vector<int32_t> got, mapping(1024), rows(1024);
  for (;;) {
    const int32_t n = probe.getOutput(1024, mapping.data(), rows.data());
    if (n == 0) {
      break;
    }
    got.insert(got.end(), mapping.begin(), mapping.begin() + n);
  }
  std::vector<int32_t> expected;
  for (int p = 0; p < kProbe; ++p) {
    if (keys.count(pk[p])) {
      expected.push_back(p);
    }
  }
  EXPECT(got == expected);
  // indices of the semi join's output over a dictionary the probe side already carried
  std::vector<int32_t> inner(kProbe), composed(got.size());
  for (int i = 0; i < kProbe; ++i) {
    inner[i] = (i * 7) % kProbe;
  }
  vx355::composeIndices(inner.data(), kProbe, got.data(), static_cast<int32_t>(got.size()), composed.data(), VX355_MEM_HOST);
  for (size_t i = 0; i < got.size(); ++i) {
    EXPECT(composed[i] == inner[got[i]]);
  }
  return 0;
}

int testJoin(vx355_join_type type) {
  // Two build drivers, duplicate and null keys, one payload column; inner / left / right / full.
  std::mt19937_64 rng(11);
  const int kBuild = 4000, kProbe = 9000;
  std::vector<int64_t> bk(kBuild), bpay(kBuild), pk(kProbe);
  std::vector<uint64_t> bNulls((kBuild + 63) / 64, ~0ULL), pNulls((kProbe + 63) / 64, ~0ULL);
  for (int i = 0; i < kBuild; ++i) {
    bk[i] = rng() % 1500;
    bpay[i] = i * 10;
    if (rng() % 20 == 0) {
      bNulls[i >> 6] &= ~(1ULL << (i & 63));
    }
  }
  for (int i = 0; i < kProbe; ++i) {
    pk[i] = int64_t(rng() % 2500) - 200;
    if (rng() % 20 == 0) {
      pNulls[i >> 6] &= ~(1ULL << (i & 63));
    }
  }
  auto valid = [](const std::vector<uint64_t>& w, int i) { return (w[i >> 6] >> (i & 63)) & 1; };
  const bool keepsNulls = type == VX355_JOIN_RIGHT || type == VX355_JOIN_FULL;
  // expected multiset of (probe row or -1, payload or -1)
  std::multiset<std::pair<int32_t, int64_t>> expected;
  std::vector<char> buildMatched(kBuild, 0);
  for (int p = 0; p < kProbe; ++p) {
    bool any = false;
    if (valid(pNulls, p)) {
      for (int b = 0; b < kBuild; ++b) {
        if (valid(bNulls, b) && bk[b] == pk[p]) {
          expected.insert({p, bpay[b]});
          buildMatched[b] = 1;
          any = true;
        }
      }
    }
    if (!any && (type == VX355_JOIN_LEFT || type == VX355_JOIN_FULL)) {
      expected.insert({p, -1});
    }
  }
  if (keepsNulls) {
    for (int b = 0; b < kBuild; ++b) {
      if (!buildMatched[b]) {
        expected.insert({-1, bpay[b]});
      }
    }
  }
  const int half = 2048;  // a multiple of 64: null bitmaps split on a word boundary
  vx355::HashBuild b1({0}, {VX355_BIGINT}, {1}, {VX355_BIGINT}, type), b2({0}, {VX355_BIGINT}, {1}, {VX355_BIGINT}, type);
  vx355_column c1[2] = {flat(VX355_BIGINT, bk.data(), bNulls.data()), flat(VX355_BIGINT, bpay.data())};
  vx355_column c2[2] = {flat(VX355_BIGINT, bk.data() + half, bNulls.data() + half / 64),
                        flat(VX355_BIGINT, bpay.data() + half)};
  b1.addInput(vx355_batch{half, 2, c1});
  b2.addInput(vx355_batch{kBuild - half, 2, c2});
  vx355::JoinTable table = b1.noMoreInput({&b2});
  EXPECT(b1.isFinished() && b2.isFinished());
  EXPECT(table.stats().has_duplicates == 1);
  vx355::HashProbe probe(table, {0}, type);
  vx355_column pc = flat(VX355_BIGINT, pk.data(), pNulls.data());
  EXPECT(probe.needsInput());
  probe.addInput(vx355_batch{kProbe, 1, &pc});
  std::multiset<std::pair<int32_t, int64_t>> got;
  const int32_t kMax = 1000;
  std::vector<int32_t> mapping(kMax), rows(kMax);
  int32_t last = -1;
  for (;;) {
    OutCol pay(VX355_BIGINT, 8, kMax);
    const int32_t id = 0;
    const int32_t n = probe.getOutput(kMax, mapping.data(), rows.data(), &pay.desc, &id, 1);
    if (n == 0) {
      break;
    }
    for (int32_t i = 0; i < n; ++i) {
      EXPECT(mapping[i] >= last);  // ascending probe rows, matches of one row contiguous
      last = mapping[i];
      EXPECT(pay.valid(i) == (rows[i] >= 0));
      got.insert({mapping[i], rows[i] >= 0 ? pay.at<int64_t>(i) : -1});
    }
  }
  probe.noMoreInput();
  EXPECT(probe.isFinished());
  if (keepsNulls) {
    for (;;) {
      OutCol pay(VX355_BIGINT, 8, kMax);
      const int32_t id = 0;
      const int32_t n = probe.getBuildSideOutput(kMax, rows.data(), &pay.desc, &id, 1);
      if (n == 0) {
        break;
      }
      for (int32_t i = 0; i < n; ++i) {
        got.insert({-1, pay.at<int64_t>(i)});
      }
    }
  }
  EXPECT(got == expected);
  return 0;
}


int testDistinctAndPages() {
  // SELECT k, count(DISTINCT v), sum(DISTINCT v) GROUP BY k (exec/DistinctAggregations.cpp) against
  // std::set, then the same rows as two PrestoPages (exec/PartitionedOutput.cpp) whose framing is
  // checked field by field.
  std::mt19937_64 rng(11);
  const int kRows = 20000;
  std::vector<int64_t> k(kRows), v(kRows);
  std::vector<uint64_t> vNulls((kRows + 63) / 64, ~0ULL);
  std::map<int64_t, std::set<int64_t>> expected;
  std::vector<int64_t> firstSeen;
  int64_t nullsInFirstPage = 0;
  const int kSplit = 7777;
  for (int i = 0; i < kRows; ++i) {
    k[i] = int64_t(rng() % 50);
    v[i] = int64_t(rng() % 20) - 5;
    if (rng() % 8 == 0) {
      vNulls[i >> 6] &= ~(1ULL << (i & 63));
      nullsInFirstPage += i < kSplit;
    }
    if (!expected.count(k[i])) {
      firstSeen.push_back(k[i]);
      expected[k[i]];
    }
    if ((vNulls[i >> 6] >> (i & 63)) & 1) {
      expected[k[i]].insert(v[i]);
    }
  }
  vx355_column cols[2] = {flat(VX355_BIGINT, k.data()), flat(VX355_BIGINT, v.data(), vNulls.data())};
  vx355_batch batch{kRows, 2, cols};
  vx355::HashAggregation op({0}, {VX355_BIGINT},
                            {{VX355_AGG_COUNT, 1, -1, VX355_BIGINT, -1, VX355_AGG_FN_DISTINCT},
                             {VX355_AGG_SUM, 1, -1, VX355_BIGINT, -1, VX355_AGG_FN_DISTINCT}});
  op.addInput(batch);
  op.noMoreInput();
  OutCol key(VX355_BIGINT, 8, 64), cnt(VX355_BIGINT, 8, 64), sum(VX355_BIGINT, 8, 64);
  vx355_out_column out[3] = {key.desc, cnt.desc, sum.desc};
  const int32_t n = op.getOutput(out, 3, 64);
  EXPECT(n == int32_t(firstSeen.size()) && op.isFinished());
  for (int32_t i = 0; i < n; ++i) {
    EXPECT(key.at<int64_t>(i) == firstSeen[i]);
    const auto& values = expected[firstSeen[i]];
    EXPECT(cnt.at<int64_t>(i) == int64_t(values.size()));
    int64_t total = 0;
    for (auto x : values) {
      total += x;
    }
    EXPECT(sum.valid(i) == !values.empty());
    EXPECT(values.empty() || sum.at<int64_t>(i) == total);
  }

  std::vector<int64_t> pageOffsets;
  const std::vector<char> pages = vx355::prestoSerialize(batch, nullptr, VX355_MEM_HOST, {0, kSplit, kRows}, 0, &pageOffsets);
  EXPECT(pageOffsets.size() == 3 && pageOffsets[2] == int64_t(pages.size()));
  auto i32 = [&](int64_t at) {
    int32_t x;
    std::memcpy(&x, pages.data() + at, 4);
    return x;
  };
  // page 0: 21-byte header, 2 columns; column 0 without nulls, column 1 with a bitmap
  EXPECT(i32(0) == kSplit && pages[4] == 0 && i32(5) == pageOffsets[1] - 21 && i32(9) == i32(5));
  EXPECT(i32(21) == 2 && i32(25) == 10 && std::memcmp(pages.data() + 29, "LONG_ARRAY", 10) == 0 && i32(39) == kSplit);
  EXPECT(pages[43] == 0);
  int64_t at = 44 + int64_t(kSplit) * 8;  // past column 0's values
  EXPECT(i32(at) == 10 && i32(at + 14) == kSplit && pages[at + 18] == 1);
  at += 19 + (kSplit + 7) / 8 + (kSplit - nullsInFirstPage) * 8;
  EXPECT(at == pageOffsets[1]);
  EXPECT(i32(pageOffsets[1]) == kRows - kSplit);
  return 0;
}

}  // namespace

// The multi-GPU orchestration a C++ host calls, at the world size one GPU
// offers: partial -> final merge through vx355_agg_merge_partials and the repartitioned join of
// BASELINE config 5 through vx355_join_repartition, checked against straight C++.
int testDistributedEntryPoints() {
  vx355::Communicator comm(vx355::Communicator::uniqueId(), 1, 0);
  EXPECT(comm.worldSize() == 1);
  std::mt19937_64 rng(23);
  const int kRows = 50000, kGroups = 37;
  std::vector<int64_t> keys(kRows), vals(kRows);
  std::map<int64_t, std::pair<int64_t, int64_t>> want;  // key -> (sum, count)
  for (int i = 0; i < kRows; ++i) {
    keys[i] = int64_t(rng() % kGroups) * 1000003;
    vals[i] = int64_t(rng() % 100000) - 50000;
    want[keys[i]].first += vals[i];
    want[keys[i]].second += 1;
  }
  vx355_column cols[2] = {flat(VX355_BIGINT, keys.data()), flat(VX355_BIGINT, vals.data())};
  vx355::HashAggregation partial({0}, {VX355_BIGINT},
                                 {{VX355_AGG_SUM, 1, -1, VX355_BIGINT, -1, 0}, {VX355_AGG_COUNT_STAR, -1, -1, VX355_BIGINT, -1, 0}},
                                 VX355_STEP_PARTIAL);
  partial.addInput(vx355_batch{kRows, 2, cols});
  partial.noMoreInput();
  const int32_t keyCol = 0, keyType = VX355_BIGINT;
  const vx355_agg_fn finalAggs[2] = {{VX355_AGG_SUM, 1, -1, VX355_BIGINT, -1, 0}, {VX355_AGG_COUNT, 2, -1, VX355_BIGINT, -1, 0}};
  vx355_agg_spec finalSpec{};
  finalSpec.num_keys = 1;
  finalSpec.key_cols = &keyCol;
  finalSpec.key_types = &keyType;
  finalSpec.num_aggs = 2;
  finalSpec.aggs = finalAggs;
  finalSpec.step = VX355_STEP_FINAL;
  vx355_agg* merged = vx355::mergePartials(comm, partial.handle(), finalSpec);
  std::vector<int64_t> outKey(kGroups), outSum(kGroups), outCount(kGroups);
  std::vector<uint64_t> valid(3, 0);
  vx355_out_column out[3] = {{VX355_BIGINT, VX355_MEM_HOST, outKey.data(), &valid[0]},
                             {VX355_BIGINT, VX355_MEM_HOST, outSum.data(), &valid[1]},
                             {VX355_BIGINT, VX355_MEM_HOST, outCount.data(), &valid[2]}};
  int32_t n = 0, finished = 0;
  vx355::check(vx355_agg_get_output(merged, out, 3, kGroups, &n, &finished));
  vx355_agg_destroy(merged);
  EXPECT(n == int32_t(want.size()) && finished == 1);
  for (int32_t i = 0; i < n; ++i) {
    EXPECT(want.count(outKey[i]) == 1 && want[outKey[i]].first == outSum[i] && want[outKey[i]].second == outCount[i]);
  }
  // repartitioned join: dim(pk unique, a) x fact(fk, m), probe side in 3 pipelined chunks
  const int kDim = 4000, kFact = 30000;
  std::vector<int64_t> pk(kDim), a(kDim), fk(kFact), m(kFact);
  std::map<int64_t, int64_t> dim;
  for (int i = 0; i < kDim; ++i) {
    pk[i] = int64_t(i) * 7919 + 11;
    a[i] = int64_t(rng() >> 20);
    dim[pk[i]] = a[i];
  }
  for (int i = 0; i < kFact; ++i) {
    fk[i] = (rng() % 10 == 0) ? -5 : pk[rng() % kDim];
    m[i] = i;
  }
  vx355_column buildCols[2] = {flat(VX355_BIGINT, pk.data()), flat(VX355_BIGINT, a.data())};
  vx355_column probeCols[2] = {flat(VX355_BIGINT, fk.data()), flat(VX355_BIGINT, m.data())};
  const int32_t zero = 0, one = 1, bigint = VX355_BIGINT;
  vx355_join_build_spec buildSpec{};
  buildSpec.num_keys = 1;
  buildSpec.key_cols = &zero;
  buildSpec.key_types = &bigint;
  buildSpec.num_dependents = 1;
  buildSpec.dependent_cols = &one;
  buildSpec.dependent_types = &bigint;
  buildSpec.join_type = VX355_JOIN_INNER;
  vx355_join_probe_spec probeSpec{};
  probeSpec.num_keys = 1;
  probeSpec.key_cols = &zero;
  probeSpec.join_type = VX355_JOIN_INNER;
  int64_t matches = 0, chunksSeen = 0;
  bool wrong = false;
  vx355::JoinTable table = vx355::repartitionedJoin(
      comm, buildSpec, vx355_batch{kDim, 2, buildCols}, probeSpec, vx355_batch{kFact, 2, probeCols}, 3,
      [&](int32_t chunk, const vx355_batch& received, vx355_join_probe* probe) {
        wrong = wrong || chunk != chunksSeen;
        ++chunksSeen;
        std::vector<int64_t> keysHere(received.num_rows);
        vx355::check(vx355_memcpy_d2h(keysHere.data(), received.cols[0].values, size_t(received.num_rows) * 8));
        std::vector<int32_t> mapping(1000), buildRows(1000);
        std::vector<int64_t> payload(1000);
        uint64_t payloadValid[16];
        vx355_out_column payloadCol{VX355_BIGINT, VX355_MEM_HOST, payload.data(), payloadValid};
        for (;;) {
          int32_t got = 0, fin = 0;
          vx355::check(vx355_join_probe_get_output(probe, 1000, mapping.data(), buildRows.data(), VX355_MEM_HOST,
                                                   &payloadCol, &zero, 1, &got, &fin));
          for (int32_t i = 0; i < got; ++i) {
            wrong = wrong || dim.at(keysHere[mapping[i]]) != payload[i];
          }
          matches += got;
          if (fin) {
            break;
          }
        }
      });
  int64_t expected = 0;
  for (int i = 0; i < kFact; ++i) {
    expected += dim.count(fk[i]);
  }
  EXPECT(!wrong && chunksSeen == 3 && matches == expected && table.stats().num_rows == kDim);
  return 0;
}

// The asynchronous boundary as a Driver would use it: queue vectors while isBlocked() allows,
// release each vector when its ticket has completed, noMoreInput drains the rest.
int testAsyncInput() {
  std::mt19937_64 rng(29);
  const int kBatches = 64, kRows = 4096, kGroups = 101;
  std::vector<std::vector<int64_t>> keys(kBatches), vals(kBatches);
  std::map<int64_t, std::pair<int64_t, int64_t>> want;
  for (int b = 0; b < kBatches; ++b) {
    keys[b].resize(kRows);
    vals[b].resize(kRows);
    for (int i = 0; i < kRows; ++i) {
      keys[b][i] = int64_t(rng() % kGroups) * 7 - 100;
      vals[b][i] = int64_t(rng() % 2000) - 1000;
      want[keys[b][i]].first += vals[b][i];
      want[keys[b][i]].second += 1;
    }
  }
  vx355::HashAggregation op({0}, {VX355_BIGINT},
                            {{VX355_AGG_SUM, 1, -1, VX355_BIGINT, -1, 0}, {VX355_AGG_COUNT_STAR, -1, -1, VX355_BIGINT, -1, 0}});
  std::vector<std::array<vx355_column, 2>> cols(kBatches);
  int64_t lastTicket = 0;
  for (int b = 0; b < kBatches; ++b) {
    while (op.isBlocked(4)) {
      std::this_thread::yield();  // a Driver would return its ContinueFuture here
    }
    cols[b] = {flat(VX355_BIGINT, keys[b].data()), flat(VX355_BIGINT, vals[b].data())};
    lastTicket = op.addInputAsync(vx355_batch{kRows, 2, cols[b].data()});
  }
  EXPECT(lastTicket == kBatches);
  std::vector<int64_t> outKey(kGroups), outSum(kGroups), outCount(kGroups);
  std::vector<uint64_t> valid(6, 0);
  vx355_out_column out[3] = {{VX355_BIGINT, VX355_MEM_HOST, outKey.data(), valid.data()},
                             {VX355_BIGINT, VX355_MEM_HOST, outSum.data(), valid.data() + 2},
                             {VX355_BIGINT, VX355_MEM_HOST, outCount.data(), valid.data() + 4}};
  // noMoreInput and the output page queue behind the batches; the callback (the library's worker thread) is what
  // a shim fulfils its ContinuePromise from
  struct Done {
    std::atomic<int> fired{0};
    int status = -1;
    int32_t rows = -1, finished = -1;
  } done;
  EXPECT(op.noMoreInputAsync() == kBatches + 1);
  const int64_t page = op.getOutputAsync(
      out, 3, kGroups,
      [](void* arg, int status, int32_t rows, int32_t finished) {
        auto* d = static_cast<Done*>(arg);
        d->status = status;
        d->rows = rows;
        d->finished = finished;
        d->fired.store(1, std::memory_order_release);
      },
      &done);
  EXPECT(page == kBatches + 2);
  while (done.fired.load(std::memory_order_acquire) == 0) {
    std::this_thread::yield();  // a Driver would be off the thread, waiting for the future
  }
  // the result may be taken as soon as the callback has fired (ABI 8: the callback runs right BEFORE the ticket
  // counts as completed, so that whoever sees completed >= ticket may free the callback's argument)
  EXPECT(done.status == VX355_OK && done.finished == 1);
  const int32_t n = op.outputResult(page);
  op.wait();
  EXPECT(op.completedTickets() == page && op.inFlight() == 0);
  EXPECT(n == done.rows && op.isFinished());
  EXPECT(n == static_cast<int32_t>(want.size()));
  for (int32_t i = 0; i < n; ++i) {
    EXPECT(want.count(outKey[i]) == 1);
    EXPECT(outSum[i] == want[outKey[i]].first && outCount[i] == want[outKey[i]].second);
  }
  return 0;
}

int main() {
  try {
    vx355::init(0);
    if (testDistributedEntryPoints()) {
      return 1;
    }
    if (testAggregation()) {
      return 1;
    }
    if (testAsyncInput()) {
      std::fprintf(stderr, "asynchronous input failed\n");
      return 1;
    }
    if (testDistinctAndPages()) {
      return 1;
    }
    if (testSemiJoinWithoutDuplicates()) {
      std::fprintf(stderr, "semi join over a build side without duplicates failed\n");
      return 1;
    }
    for (auto t : {VX355_JOIN_INNER, VX355_JOIN_LEFT, VX355_JOIN_RIGHT, VX355_JOIN_FULL}) {
      if (testJoin(t)) {
        std::fprintf(stderr, "join type %d failed\n", int(t));
        return 1;
      }
    }
  } catch (const vx355::Error& e) {
    std::fprintf(stderr, "vx355 error %d: %s\n", e.status, e.what());
    return 2;
  }
  std::printf("operator_test: aggregation and inner/left/right/full joins match the expected results\n");
  return 0;
}
