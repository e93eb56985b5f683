// TEST (GPU): the Velox-side adapter (shim/*.cpp, compiled against tests/velox_api_stub) driving
// libvx355 on a real device, the way a Velox Driver would: registerVx355(), the DriverAdapter
// replaces the operators of each pipeline, RowVectors of 10 000 rows go in through
// Operator::addInput and come out of Operator::getOutput.
//   1. the reference's TPC-H Q1 plan (exec/tests/utils/TpchQueryBuilder.cpp:203-252): [FilterProject,
//      partial HashAggregation] fused into one operator -> RowVectors with ROW(DOUBLE, BIGINT) avg
//      intermediates -> final HashAggregation; and the same with the FilterProject left on the CPU
//      side (unfused), with a tiny max_partial_aggregation_memory so that the flush path runs.
//   2. a two-pipeline inner join with build payload through Vx355HashBuild / Vx355HashProbe and the
//      table rendezvous; with a scan stand-in in front of the probe that accepts dynamic filters
//      (HashProbe::pushdownDynamicFilters, exec/HashProbe.cpp:408-457): a value filter for 50 000 build
//      keys, the table's Bloom blocks for 150 000.
//   3. the runtime stats of every operator: the reference's hashtable.* names (exec/HashTable.h:155-163),
//      flushTimes, dynamicFiltersProduced / Accepted, and the gpu.* counters of SURVEY.md section 5.
// Expected values come from plain loops over the same host data (all arithmetic exact in DOUBLE:
// prices are multiples of 1/4, discounts and taxes multiples of 1/4).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <tuple>

#include "Vx355Adapter.h"
#include "Vx355JoinAdapter.h"
#include "shim_test_plans.h"

using namespace facebook::velox;
using namespace facebook::velox::vx355;

#define EXPECT(cond)                                                                 \
  do {                                                                               \
    if (!(cond)) {                                                                   \
      std::fprintf(stderr, "%s:%d: EXPECT(%s) failed\n", __FILE__, __LINE__, #cond); \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

namespace {

memory::MemoryPool gPool;

struct Lineitem {
  std::vector<RowVectorPtr> batches;
  // expected per (returnflag, linestatus): sums, counts (over rows passing the filter)
  struct Group {
    double qty{0}, price{0}, disc{0}, charge{0}, discount{0};
    int64_t count{0};
  };
  std::map<std::pair<std::string, std::string>, Group> expected;
};

Lineitem makeLineitem(int64_t rows, int32_t batchRows, const RowTypePtr& type) {
  Lineitem out;
  std::mt19937_64 rng(42);
  const char* flags[3] = {"A", "N", "R"};
  const char* statuses[2] = {"F", "O"};
  for (int64_t begin = 0; begin < rows; begin += batchRows) {
    const auto n = static_cast<vector_size_t>(std::min<int64_t>(batchRows, rows - begin));
    auto batch = std::static_pointer_cast<RowVector>(BaseVector::create(type, n, &gPool));
    for (vector_size_t i = 0; i < n; ++i) {
      const std::string flag = flags[rng() % 3], status = statuses[rng() % 2];
      const double qty = 1 + static_cast<double>(rng() % 50);
      const double price = static_cast<double>(rng() % 400000) / 4;
      const double discount = static_cast<double>(rng() % 3) / 4;
      const double tax = static_cast<double>(rng() % 3) / 4;
      const int32_t shipdate = 10300 + static_cast<int32_t>(rng() % 300);
      batch->childAt(0)->asFlatVector<StringView>()->set(i, StringView(flag));
      batch->childAt(1)->asFlatVector<StringView>()->set(i, StringView(status));
      batch->childAt(2)->asFlatVector<double>()->set(i, qty);
      batch->childAt(3)->asFlatVector<double>()->set(i, price);
      batch->childAt(4)->asFlatVector<double>()->set(i, discount);
      batch->childAt(5)->asFlatVector<double>()->set(i, tax);
      batch->childAt(6)->asFlatVector<int32_t>()->set(i, shipdate);
      if (shipdate <= shimtest::kQ1Cutoff) {
        auto& g = out.expected[{flag, status}];
        g.qty += qty;
        g.price += price;
        g.disc += price * (1.0 - discount);
        g.charge += price * (1.0 - discount) * (1.0 + tax);
        g.discount += discount;
        ++g.count;
      }
    }
    out.batches.push_back(batch);
  }
  return out;
}

// The Driver's loop for one pipeline whose source is 'inputs' (exec/Driver.cpp:538-800 reduced to
// its contract: isBlocked -> needsInput -> addInput, getOutput until null, noMoreInput, drain).
std::vector<RowVectorPtr> runPipeline(exec::Driver& driver, const std::vector<RowVectorPtr>& inputs) {
  std::vector<RowVectorPtr> outputs;
  auto operators = driver.operators();
  auto pump = [&](size_t from, RowVectorPtr batch, auto&& self) -> void {
    // hands 'batch' to operator 'from' and moves whatever comes out down the pipeline
    if (from >= operators.size()) {
      outputs.push_back(std::move(batch));
      return;
    }
    auto* op = operators[from];
    ContinueFuture future = ContinueFuture::makeEmpty();
    while (op->isBlocked(&future) != exec::BlockingReason::kNotBlocked) {
      future.wait();
    }
    while (!op->needsInput()) {
      auto out = op->getOutput();
      if (out != nullptr) {
        self(from + 1, out, self);
      }
    }
    op->addInput(std::move(batch));
    while (auto out = op->getOutput()) {
      self(from + 1, out, self);
    }
  };
  for (const auto& input : inputs) {
    pump(0, input, pump);
  }
  for (size_t i = 0; i < operators.size(); ++i) {
    auto* op = operators[i];
    op->noMoreInput();
    while (!op->isFinished()) {
      ContinueFuture future = ContinueFuture::makeEmpty();
      if (op->isBlocked(&future) != exec::BlockingReason::kNotBlocked) {
        future.wait();
        continue;
      }
      auto out = op->getOutput();
      if (out != nullptr) {
        pump(i + 1, out, pump);
      }
    }
  }
  for (auto* op : operators) {
    op->close();
  }
  return outputs;
}

std::shared_ptr<exec::Driver> newDriver(const std::shared_ptr<exec::Task>& task, int pipeline) {
  return std::make_shared<exec::Driver>(std::make_unique<exec::DriverCtx>(task, 0, pipeline, 0, 0));
}

/// The CPU FilterProject of the unfused variant, evaluated by the test itself (the stub has no
/// expression evaluator): filter + the seven projected columns of the Q1 plan, flat.
class TestFilterProject : public exec::Operator {
 public:
  TestFilterProject(int32_t id, exec::DriverCtx* ctx, const std::shared_ptr<const core::ProjectNode>& project)
      : Operator(ctx, project->outputType(), id, project->id(), "TestFilterProject") {}
  bool needsInput() const override {
    return input_ == nullptr;
  }
  void addInput(RowVectorPtr input) override {
    input_ = std::move(input);
  }
  RowVectorPtr getOutput() override {
    if (input_ == nullptr) {
      return nullptr;
    }
    auto in = std::move(input_);
    input_ = nullptr;
    std::vector<vector_size_t> selected;
    auto* ship = in->childAt(6)->asFlatVector<int32_t>();
    for (vector_size_t i = 0; i < in->size(); ++i) {
      if (ship->valueAt(i) <= shimtest::kQ1Cutoff) {
        selected.push_back(i);
      }
    }
    if (selected.empty()) {
      return nullptr;
    }
    const auto n = static_cast<vector_size_t>(selected.size());
    // pass-through columns behind a dictionary (what FilterProject does, exec/OperatorUtils.cpp:393-422),
    // computed columns flat
    auto indices = allocateIndices(n, pool());
    std::memcpy(indices->asMutable<vector_size_t>(), selected.data(), n * sizeof(vector_size_t));
    auto disc = BaseVector::create<FlatVector<double>>(DOUBLE(), n, pool());
    auto charge = BaseVector::create<FlatVector<double>>(DOUBLE(), n, pool());
    for (vector_size_t j = 0; j < n; ++j) {
      const auto i = selected[j];
      const double price = in->childAt(3)->asFlatVector<double>()->valueAt(i);
      const double d = in->childAt(4)->asFlatVector<double>()->valueAt(i);
      const double t = in->childAt(5)->asFlatVector<double>()->valueAt(i);
      disc->set(j, price * (1.0 - d));
      charge->set(j, price * (1.0 - d) * (1.0 + t));
    }
    auto wrap = [&](int c) { return exec::wrapChild(n, indices, in->childAt(c)); };
    std::vector<VectorPtr> children = {wrap(0), wrap(1), wrap(2), wrap(3), disc, charge, wrap(4)};
    return std::make_shared<RowVector>(pool(), outputType_, nullptr, n, std::move(children));
  }
  exec::BlockingReason isBlocked(ContinueFuture*) override {
    return exec::BlockingReason::kNotBlocked;
  }
  bool isFinished() override {
    return noMoreInput_ && input_ == nullptr;
  }
};

/// A TableScan stand-in at the head of the probe pipeline: passes its input on, accepts dynamic filters
/// (exec/Operator.h:315-329; TableScan::addDynamicFilterLocked hands them to the connector's reader) and
/// applies them to the rows it passes on, as a scan would.
class TestScan : public exec::Operator {
 public:
  TestScan(int32_t id, exec::DriverCtx* ctx, const std::shared_ptr<const core::ValuesNode>& node)
      : Operator(ctx, node->outputType(), id, node->id(), "TestScan") {}
  bool canAddDynamicFilter() const override {
    return true;
  }
  void addDynamicFilterLocked(const core::PlanNodeId& producer, const exec::PushdownFilters& filters) override {
    producer_ = producer;
    for (const auto& [channel, filter] : filters.filters) {
      filters_[channel] = filter;
    }
  }
  bool needsInput() const override {
    return input_ == nullptr;
  }
  void addInput(RowVectorPtr input) override {
    input_ = std::move(input);
  }
  RowVectorPtr getOutput() override {
    if (input_ == nullptr) {
      return nullptr;
    }
    auto in = std::move(input_);
    input_ = nullptr;
    if (filters_.empty()) {
      return in;
    }
    std::vector<vector_size_t> selected;
    for (vector_size_t i = 0; i < in->size(); ++i) {
      bool pass = true;
      for (const auto& [channel, filter] : filters_) {
        pass = pass && filter->testInt64(in->childAt(channel)->asFlatVector<int64_t>()->valueAt(i));
      }
      if (pass) {
        selected.push_back(i);
      }
    }
    rowsDropped += in->size() - static_cast<int64_t>(selected.size());
    if (selected.empty()) {
      return nullptr;
    }
    const auto n = static_cast<vector_size_t>(selected.size());
    auto out = std::static_pointer_cast<RowVector>(BaseVector::create(outputType_, n, pool()));
    for (vector_size_t j = 0; j < n; ++j) {
      out->childAt(0)->asFlatVector<int64_t>()->set(j, in->childAt(0)->asFlatVector<int64_t>()->valueAt(selected[j]));
      out->childAt(1)->asFlatVector<double>()->set(j, in->childAt(1)->asFlatVector<double>()->valueAt(selected[j]));
      out->childAt(2)->asFlatVector<int32_t>()->set(j, in->childAt(2)->asFlatVector<int32_t>()->valueAt(selected[j]));
    }
    return out;
  }
  exec::BlockingReason isBlocked(ContinueFuture*) override {
    return exec::BlockingReason::kNotBlocked;
  }
  bool isFinished() override {
    return noMoreInput_ && input_ == nullptr;
  }
  std::map<column_index_t, common::FilterPtr> filters_;
  core::PlanNodeId producer_;
  int64_t rowsDropped{0};
};

int64_t statSum(const exec::Operator& op, const std::string& name) {
  const auto stats = op.statsCopy();
  const auto it = stats.runtimeStats.find(name);
  return it == stats.runtimeStats.end() ? -1 : it->second.sum;
}

int checkQ1(const std::vector<RowVectorPtr>& finals, const Lineitem& data) {
  std::map<std::pair<std::string, std::string>, int> seen;
  for (const auto& page : finals) {
    EXPECT(page->childrenSize() == 10);
    for (vector_size_t r = 0; r < page->size(); ++r) {
      const auto key = std::make_pair(page->childAt(0)->asFlatVector<StringView>()->valueAt(r).str(),
                                      page->childAt(1)->asFlatVector<StringView>()->valueAt(r).str());
      EXPECT(data.expected.count(key) == 1);
      EXPECT(++seen[key] == 1);
      const auto& g = data.expected.at(key);
      auto d = [&](int c) { return page->childAt(c)->asFlatVector<double>()->valueAt(r); };
      EXPECT(d(2) == g.qty && d(3) == g.price && d(4) == g.disc && d(5) == g.charge);
      EXPECT(d(6) == g.qty / g.count && d(7) == g.price / g.count && d(8) == g.discount / g.count);
      EXPECT(page->childAt(9)->asFlatVector<int64_t>()->valueAt(r) == g.count);
      for (int c = 0; c < 10; ++c) {
        EXPECT(!page->childAt(c)->isNullAt(r));
      }
    }
  }
  EXPECT(seen.size() == data.expected.size());
  return 0;
}

int testQ1(bool fusedVariant, uint64_t maxPartialMemory, int* flushes) {
  auto plan = shimtest::q1Plan(true);
  auto data = makeLineitem(400000, 10000, plan.scan->outputType());
  auto task = std::make_shared<exec::Task>(fusedVariant ? "q1_fused" : "q1_unfused");
  task->mutableQueryConfig().maxPartialAggregationMemory = maxPartialMemory;
  task->mutableQueryConfig().preferredBatchRows = 1024;

  // pipeline 0: scan -> [filter -> project] -> partial aggregation
  auto d0 = newDriver(task, 0);
  exec::DriverFactory f0;
  f0.planNodes = {plan.scan, plan.filter, plan.project, plan.partial};
  if (fusedVariant) {
    d0->mutableOperators().push_back(std::make_unique<exec::FilterProject>(0, d0->driverCtx(), plan.filter, plan.project));
  } else {
    d0->mutableOperators().push_back(std::make_unique<TestFilterProject>(0, d0->driverCtx(), plan.project));
  }
  d0->mutableOperators().push_back(std::make_unique<exec::HashAggregation>(1, d0->driverCtx(), plan.partial));
  EXPECT(adaptDriver(f0, *d0));
  auto ops0 = d0->operators();
  if (fusedVariant) {
    EXPECT(ops0.size() == 1 && ops0[0]->operatorType() == "Vx355HashAggregation" && ops0[0]->operatorId() == 0);
  } else {
    EXPECT(ops0.size() == 2 && ops0[1]->operatorType() == "Vx355HashAggregation");
  }
  auto partials = runPipeline(*d0, data.batches);
  EXPECT(!partials.empty());
  *flushes = static_cast<int>(partials.size());
  {
    // runtime stats of the partial aggregation (recorded when it finished): the reference's names and the gpu.* ones
    const auto& agg = *ops0.back();
    EXPECT(statSum(agg, "hashtable.capacity") > 0 && statSum(agg, "hashtable.numRehashes") >= 0);
    EXPECT(statSum(agg, "hashtable.hashMode") >= 0 && statSum(agg, "hashtable.numDistinct") >= 0);
    EXPECT(statSum(agg, "gpu.kernelNanos") > 0 && statSum(agg, "gpu.kernelLaunches") > 0);
    EXPECT(statSum(agg, "gpu.h2dBytes") >= 400000 * 8 && statSum(agg, "gpu.hbmBytesRead") >= statSum(agg, "gpu.h2dBytes"));
    EXPECT(statSum(agg, "gpu.d2hBytes") > 0 && statSum(agg, "gpu.hbmBytesWritten") >= statSum(agg, "gpu.d2hBytes"));
    EXPECT((statSum(agg, "flushTimes") >= 1) == (maxPartialMemory == 1));
  }
  for (const auto& page : partials) {
    // the partial step's output type: avg travels as ROW(DOUBLE sum, BIGINT count)
    EXPECT(page->childrenSize() == 10 && page->childAt(6)->type()->isRow());
    auto* avg = page->childAt(6)->as<RowVector>();
    EXPECT(avg != nullptr && avg->childrenSize() == 2 && avg->size() == page->size());
  }

  // pipeline 1: (local exchange) -> final aggregation
  auto d1 = newDriver(task, 1);
  exec::DriverFactory f1;
  f1.planNodes = {plan.final};
  d1->mutableOperators().push_back(std::make_unique<exec::HashAggregation>(0, d1->driverCtx(), plan.final));
  EXPECT(adaptDriver(f1, *d1));
  EXPECT(d1->operators()[0]->operatorType() == "Vx355HashAggregation");
  auto finals = runPipeline(*d1, partials);
  return checkQ1(finals, data);
}

int testJoin(bool withFilter, bool withScan = false, int64_t numOrders = 50000) {
  // orders (build): o_orderkey, o_orderdate, o_shippriority; lineitem (probe): l_orderkey, l_extendedprice, l_shipdate.
  // withFilter: TPC-H Q3's shape - FilterProject(l_shipdate > DATE) in front of the probe (a FilterNode alone), which
  // the adapter folds into the probe operator (vx355_join_probe_set_input_filter).
  constexpr int32_t kDate = 9204;
  auto probeNode = std::make_shared<core::ValuesNode>("l", ROW({"l_orderkey", "l_extendedprice", "l_shipdate"}, {BIGINT(), DOUBLE(), DATE()}));
  std::shared_ptr<const core::FilterNode> filterNode;
  core::PlanNodePtr probeSource = probeNode;
  if (withFilter) {
    filterNode = std::make_shared<core::FilterNode>(
        "lf", shimtest::call("gt", BOOLEAN(), {shimtest::field(probeNode->outputType(), "l_shipdate"),
                                               shimtest::constant(DATE(), Variant(kDate))}), probeNode);
    probeSource = filterNode;
  }
  auto buildNode = std::make_shared<core::ValuesNode>("o", ROW({"o_orderkey", "o_orderdate", "o_shippriority"}, {BIGINT(), DATE(), INTEGER()}));
  auto outputType = ROW({"l_extendedprice", "o_orderdate", "o_shippriority", "l_orderkey"}, {DOUBLE(), DATE(), INTEGER(), BIGINT()});
  auto join = std::make_shared<core::HashJoinNode>(
      "join", core::JoinType::kInner, false, false,
      std::vector<core::FieldAccessTypedExprPtr>{shimtest::field(probeNode->outputType(), "l_orderkey")},
      std::vector<core::FieldAccessTypedExprPtr>{shimtest::field(buildNode->outputType(), "o_orderkey")}, nullptr, probeSource, buildNode,
      outputType);
  std::mt19937_64 rng(7);
  const int64_t numLineitems = 300000;
  std::map<int64_t, std::pair<int32_t, int32_t>> orders;
  std::vector<RowVectorPtr> buildBatches, probeBatches;
  for (int64_t begin = 0; begin < numOrders; begin += 10000) {
    auto batch = std::static_pointer_cast<RowVector>(BaseVector::create(buildNode->outputType(), 10000, &gPool));
    for (vector_size_t i = 0; i < 10000; ++i) {
      const int64_t key = (begin + i) * 4 + 1;  // unique, a quarter of the probe key space
      const auto date = static_cast<int32_t>(9000 + rng() % 1000);
      const auto prio = static_cast<int32_t>(rng() % 5);
      batch->childAt(0)->asFlatVector<int64_t>()->set(i, key);
      batch->childAt(1)->asFlatVector<int32_t>()->set(i, date);
      batch->childAt(2)->asFlatVector<int32_t>()->set(i, prio);
      orders[key] = {date, prio};
    }
    buildBatches.push_back(batch);
  }
  int64_t expectedRows = 0;
  for (int64_t begin = 0; begin < numLineitems; begin += 10000) {
    auto batch = std::static_pointer_cast<RowVector>(BaseVector::create(probeNode->outputType(), 10000, &gPool));
    for (vector_size_t i = 0; i < 10000; ++i) {
      const int64_t key = static_cast<int64_t>(rng() % (numOrders * 4 + 8));
      const auto ship = static_cast<int32_t>(9000 + rng() % 400);
      batch->childAt(0)->asFlatVector<int64_t>()->set(i, key);
      // the price encodes the ship date, so that the output can be checked against the filter
      batch->childAt(1)->asFlatVector<double>()->set(i, static_cast<double>(ship));
      batch->childAt(2)->asFlatVector<int32_t>()->set(i, ship);
      expectedRows += (!withFilter || ship > kDate) ? orders.count(key) : 0;
    }
    probeBatches.push_back(batch);
  }
  auto task = std::make_shared<exec::Task>(std::string(withFilter ? "join_task_filter" : "join_task") + (withScan ? "_scan" : "") +
                                           std::to_string(numOrders));
  task->mutableQueryConfig().preferredBatchRows = 4096;
  // build pipeline ends in the join node (its consumer)
  auto db = newDriver(task, 1);
  exec::DriverFactory fb;
  fb.planNodes = {buildNode};
  fb.consumerNode = join;
  db->mutableOperators().push_back(std::make_unique<exec::HashBuild>(0, db->driverCtx(), join));
  auto dp = newDriver(task, 0);
  exec::DriverFactory fp;
  int32_t next = 0;
  if (withScan) {
    dp->mutableOperators().push_back(std::make_unique<TestScan>(next++, dp->driverCtx(), probeNode));
  }
  if (withFilter) {
    fp.planNodes = {probeNode, filterNode, join};
    dp->mutableOperators().push_back(std::make_unique<exec::FilterProject>(next++, dp->driverCtx(), filterNode, nullptr));
    dp->mutableOperators().push_back(std::make_unique<exec::HashProbe>(next++, dp->driverCtx(), join));
  } else {
    fp.planNodes = {probeNode, join};
    dp->mutableOperators().push_back(std::make_unique<exec::HashProbe>(next++, dp->driverCtx(), join));
  }
  // (Velox creates the Drivers of every pipeline before any of them runs)
  EXPECT(adaptDriver(fp, *dp));
  EXPECT(adaptDriver(fb, *db));
  // (with the filter: FilterProject and HashProbe became ONE operator)
  const size_t probeAt = withScan ? 1 : 0;
  EXPECT(dp->operators().size() == probeAt + 1 && dp->operators()[probeAt]->operatorId() == static_cast<int32_t>(probeAt));
  EXPECT(dp->operators()[probeAt]->operatorType() == "Vx355HashProbe" && db->operators()[0]->operatorType() == "Vx355HashBuild");
  // the probe is blocked until the table is published
  ContinueFuture waitForBuild = ContinueFuture::makeEmpty();
  EXPECT(dp->operators()[probeAt]->isBlocked(&waitForBuild) == exec::BlockingReason::kWaitForJoinBuild);
  EXPECT(waitForBuild.valid() && !waitForBuild.isReady());
  auto none = runPipeline(*db, buildBatches);
  EXPECT(none.empty());
  EXPECT(waitForBuild.isReady());
  {
    // HashBuild::addRuntimeStats: the table under the reference's names, the build's wall time, the gpu.* counters
    const auto& build = *db->operators()[0];
    EXPECT(statSum(build, "hashtable.numDistinct") == numOrders && statSum(build, "hashtable.capacity") >= numOrders);
    EXPECT(statSum(build, "hashtable.hashMode") >= 0 && statSum(build, "hashtable.buildWallNanos") > 0);
    EXPECT(statSum(build, "gpu.kernelNanos") > 0 && statSum(build, "gpu.h2dBytes") >= numOrders * 16);
  }
  auto pages = runPipeline(*dp, probeBatches);
  {
    const auto& probe = *dp->operators()[probeAt];
    // (behind a scan that applies the dynamic filter the probe sees the surviving rows only)
    EXPECT(statSum(probe, "gpu.kernelNanos") > 0 && statSum(probe, "gpu.hbmBytesRead") >= (withScan ? 8 : numLineitems * 8));
    if (withScan) {
      // the key's filter reached the scan: made from the table (<= 100 000 distinct values: the value list through
      // common::createBigintValues; more: the table's Bloom blocks), counted on both operators, applied by the scan
      auto* scan = dynamic_cast<TestScan*>(dp->operators()[0]);
      EXPECT(scan != nullptr && scan->filters_.size() == 1 && scan->filters_.count(0) == 1 && scan->producer_ == "join");
      EXPECT(statSum(probe, "dynamicFiltersProduced") == 1 && statSum(*scan, "dynamicFiltersAccepted") == 1);
      const auto& filter = scan->filters_.at(0);
      if (numOrders <= 100000) {
        EXPECT(filter->kind() == common::FilterKind::kBigintValuesUsingHashTable);
        auto* values = dynamic_cast<const common::BigintValuesUsingHashTable*>(filter.get());
        EXPECT(values != nullptr && static_cast<int64_t>(values->values().size()) == numOrders);
        EXPECT(values->min() == 1 && values->max() == (numOrders - 1) * 4 + 1);
        EXPECT(filter->testInt64(5) && !filter->testInt64(6));
        EXPECT(scan->rowsDropped > numLineitems / 2);   // three probe keys in four have no order
      } else {
        EXPECT(filter->kind() == common::FilterKind::kBigintValuesUsingBloomFilter);
        auto* bloom = dynamic_cast<common::BigintValuesUsingBloomFilter*>(filter.get());
        EXPECT(bloom != nullptr && bloom->blocksByteSize() == statSum(probe, "bloomFilterSize"));
        // the blocks hold every build key and reject most others (checked with the library's own tester)
        std::vector<int64_t> present, absent;
        for (int64_t i = 0; i < 1000; ++i) {
          present.push_back(i * 4 + 1);
          absent.push_back(i * 4 + 2);
        }
        for (const auto* keys : {&present, &absent}) {
          vx355_column column{};
          column.type_kind = VX355_BIGINT;
          column.encoding = VX355_FLAT;
          column.values = keys->data();
          column.mem = VX355_MEM_HOST;
          std::vector<uint64_t> passed(16, 0);
          EXPECT(vx355_bloom_test(reinterpret_cast<const uint32_t*>(bloom->mutableBlocks()), bloom->numBlocksHeld(), 8, &column,
                                  1000, nullptr, passed.data(), VX355_MEM_HOST) == VX355_OK);
          int64_t hits = 0;
          for (uint64_t word : passed) {
            hits += __builtin_popcountll(word);
          }
          EXPECT(keys == &present ? hits == 1000 : hits < 100);
        }
      }
    }
  }
  int64_t rows = 0;
  for (const auto& page : pages) {
    EXPECT(page->childrenSize() == 4);
    // probe columns come back wrapped in the mapping; read them through DecodedVector
    DecodedVector price(*page->childAt(0)), key(*page->childAt(3));
    for (vector_size_t r = 0; r < page->size(); ++r) {
      const int64_t k = key.data<int64_t>()[key.index(r)];
      EXPECT(orders.count(k) == 1);
      EXPECT(page->childAt(1)->asFlatVector<int32_t>()->valueAt(r) == orders[k].first);
      EXPECT(page->childAt(2)->asFlatVector<int32_t>()->valueAt(r) == orders[k].second);
      EXPECT(!withFilter || price.data<double>()[price.index(r)] > kDate);   // only rows that pass the fused filter
    }
    rows += page->size();
  }
  EXPECT(rows == expectedRows && rows > 0);
  return 0;
}

}  // namespace

int main() {
  try {
    registerVx355(0);
    EXPECT(exec::DriverFactory::adapters().size() == 1 && exec::DriverFactory::adapters()[0].label == "vx355");
    int flushes = 0;
    if (testQ1(/*fused=*/true, 1UL << 24, &flushes) != 0) {
      return 1;
    }
    std::printf("ok: Q1 plan, FilterProject fused into the aggregation (%d partial page(s))\n", flushes);
    if (testQ1(/*fused=*/false, 1UL << 24, &flushes) != 0) {
      return 1;
    }
    std::printf("ok: Q1 plan, aggregation behind a CPU FilterProject (%d partial page(s))\n", flushes);
    // max_partial_aggregation_memory of one byte: the partial operator flushes as soon as a batch has
    // completed, again and again; the final step still sees every group exactly once
    if (testQ1(/*fused=*/true, 1, &flushes) != 0) {
      return 1;
    }
    EXPECT(flushes >= 2);
    std::printf("ok: Q1 plan with partial flushes (%d partial pages)\n", flushes);
    if (testJoin(false) != 0) {
      return 1;
    }
    std::printf("ok: inner join through Vx355HashBuild / Vx355HashProbe\n");
    if (testJoin(true) != 0) {
      return 1;
    }
    std::printf("ok: FilterProject(l_shipdate > d) folded into Vx355HashProbe\n");
    if (testJoin(false, /*withScan=*/true) != 0 || testJoin(true, /*withScan=*/true) != 0) {
      return 1;
    }
    std::printf("ok: dynamic filter (value list) pushed from the join table to the probe side's scan\n");
    if (testJoin(false, /*withScan=*/true, 150000) != 0) {
      return 1;
    }
    std::printf("ok: dynamic filter (Bloom blocks of the table) pushed to the probe side's scan\n");
    // Beyond the operators' share of HBM: the build fails with the reference's memory-cap error, its handles
    // are destroyed on the way out, and the same plan runs again once the limit is lifted.
    int64_t inUse = 0;
    EXPECT(vx355_memory_usage(&inUse, nullptr, nullptr) == VX355_OK);
    EXPECT(vx355_set_memory_limit(inUse + (256 << 10)) == VX355_OK);
    bool capped = false;
    try {
      (void)testJoin(false, false, 400000);
    } catch (const VeloxRuntimeError& e) {
      capped = e.errorCode() == error_code::kMemCapExceeded && std::string(e.what()).find("memory limit") != std::string::npos;
    }
    EXPECT(capped);
    int64_t after = 0;
    EXPECT(vx355_memory_usage(&after, nullptr, nullptr) == VX355_OK && after == inUse);
    EXPECT(vx355_set_memory_limit(0) == VX355_OK);
    if (testJoin(false, false, 400000) != 0) {
      return 1;
    }
    std::printf("ok: VX355_ENOMEM -> VELOX_MEM_POOL_CAP_EXCEEDED, nothing leaked, the plan runs once the limit is lifted\n");
  } catch (const std::exception& e) {
    std::fprintf(stderr, "FAILED: %s\n", e.what());
    return 1;
  }
  std::printf("shim operator tests: results match the expected values\n");
  return 0;
}
