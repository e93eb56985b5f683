// TEST INFRASTRUCTURE ONLY — CPU oracle, see oracle.h.
// Restatement of exec/HashTable.{h,cpp} + the parts of exec/RowContainer.{h,cpp}
// the group-by and join paths need: 16-slot tagged buckets, the three hash
// modes, group probe / insert, join build (duplicate chains) and join probe.
#pragma once
#include <algorithm>
#include <atomic>
#include <thread>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "vector_hasher.h"

namespace orc {

enum class HashMode { kHash = 0, kArray = 1, kNormalizedKey = 2 };  // BaseHashTable::HashMode

struct UserError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// Row-wise arena (exec/RowContainer.cpp:140-277, Appendix B of SURVEY.md),
// simplified: every key / dependent occupies an 8-byte slot (16 for
// StringView / Timestamp), flags are one byte each. Each row is preceded by
// the 8-byte normalized key (RowContainer.h:71,755).
class RowContainer {
 public:
  struct Column {
    int32_t kind;
    int32_t offset;
    int32_t nullOffset;  // byte that is 1 when the value is null
    int32_t width;
  };

  RowContainer(const std::vector<int32_t>& keyKinds, const std::vector<int32_t>& accBytes,
               const std::vector<int32_t>& depKinds, bool hasNext) {
    int32_t off = 0;
    auto addCol = [&](int32_t kind, std::vector<Column>& out) {
      int32_t w = kindWidth(kind) == 16 ? 16 : 8;
      out.push_back({kind, off, 0, w});
      off += w;
    };
    for (auto k : keyKinds) {
      addCol(k, keys_);
    }
    for (auto k : depKinds) {
      addCol(k, deps_);
    }
    for (auto b : accBytes) {
      accOffsets_.push_back(off);
      off += b;
    }
    if (hasNext) {
      nextOffset_ = off;
      off += 8;
    }
    rowIdOffset_ = off;
    off += 8;
    for (auto& c : keys_) {
      c.nullOffset = off++;
    }
    for (auto& c : deps_) {
      c.nullOffset = off++;
    }
    for (size_t i = 0; i < accBytes.size(); ++i) {
      accNullOffsets_.push_back(off++);
    }
    rowSize_ = (off + 7) & ~7;
  }

  char* newRow() {  // RowContainer::newRow (RowContainer.cpp:282-303)
    const int32_t stride = rowSize_ + 8;
    if (chunkUsed_ + stride > chunkSize_) {
      chunkSize_ = std::max<int64_t>(stride * 1024L, 1 << 20);
      chunks_.emplace_back(new char[chunkSize_]);
      chunkUsed_ = 0;
    }
    char* row = chunks_.back().get() + chunkUsed_ + 8;
    chunkUsed_ += stride;
    std::memset(row - 8, 0, stride);
    *reinterpret_cast<int64_t*>(row + rowIdOffset_) = static_cast<int64_t>(rows_.size());
    rows_.push_back(row);
    return row;
  }

  static uint64_t& normalizedKey(char* row) { return *reinterpret_cast<uint64_t*>(row - 8); }

  // RowContainer::store (RowContainer.cpp:521-583).
  void store(const Decoded& d, int32_t index, char* row, const Column& col) {
    if (d.isNull(index)) {
      row[col.nullOffset] = 1;
      return;
    }
    uint8_t tmp;
    const void* p = d.valuePtr(index, &tmp);
    switch (col.kind) {
      case VX355_VARCHAR:
      case VX355_VARBINARY: {
        StringView sv;
        std::memcpy(&sv, p, 16);
        if (!sv.isInline()) {
          strings_.emplace_back(new char[sv.size]);
          std::memcpy(strings_.back().get(), sv.value.data, sv.size);
          sv.value.data = strings_.back().get();
        }
        std::memcpy(row + col.offset, &sv, 16);
        break;
      }
      case VX355_TIMESTAMP:
        std::memcpy(row + col.offset, p, 16);
        break;
      case VX355_REAL: {
        uint64_t bits = 0;
        std::memcpy(&bits, p, 4);
        std::memcpy(row + col.offset, &bits, 8);
        break;
      }
      case VX355_DOUBLE:
        std::memcpy(row + col.offset, p, 8);
        break;
      default: {
        int64_t v = d.int64At(index);
        std::memcpy(row + col.offset, &v, 8);
      }
    }
  }

  // RowContainer::equals / compare (RowContainer.h:1041-1077): nulls equal
  // nulls; floating point uses NaN-aware equality (NaN == NaN, 0.0 == -0.0).
  static bool equals(const char* row, const Column& col, const Decoded& d, int32_t index) {
    bool rowNull = row[col.nullOffset] != 0;
    bool vecNull = d.isNull(index);
    if (rowNull || vecNull) {
      return rowNull == vecNull;
    }
    uint8_t tmp;
    const void* p = d.valuePtr(index, &tmp);
    switch (col.kind) {
      case VX355_VARCHAR:
      case VX355_VARBINARY: {
        StringView a, b;
        std::memcpy(&a, row + col.offset, 16);
        std::memcpy(&b, p, 16);
        return a.size == b.size && std::memcmp(a.data(), b.data(), a.size) == 0;
      }
      case VX355_TIMESTAMP:
        return std::memcmp(row + col.offset, p, 16) == 0;
      case VX355_REAL: {
        float a, b;
        std::memcpy(&a, row + col.offset, 4);
        std::memcpy(&b, p, 4);
        return (std::isnan(a) && std::isnan(b)) || a == b;
      }
      case VX355_DOUBLE: {
        double a, b;
        std::memcpy(&a, row + col.offset, 8);
        std::memcpy(&b, p, 8);
        return (std::isnan(a) && std::isnan(b)) || a == b;
      }
      default: {
        int64_t a;
        std::memcpy(&a, row + col.offset, 8);
        return a == d.int64At(index);
      }
    }
  }

  // RowContainer::hash (RowContainer.cpp:927-967) for one stored column.
  static uint64_t hashStored(const char* row, const Column& col) {
    if (row[col.nullOffset]) {
      return kNullHash;
    }
    const char* p = row + col.offset;
    switch (col.kind) {
      case VX355_BOOLEAN: {
        int64_t v;
        std::memcpy(&v, p, 8);
        return hashBool(v != 0);
      }
      case VX355_TINYINT:
      case VX355_SMALLINT:
      case VX355_INTEGER: {
        int64_t v;
        std::memcpy(&v, p, 8);
        return hashInt32Like(static_cast<int32_t>(v));
      }
      default:
        return hashValue(col.kind, p);
    }
  }

  const std::vector<Column>& keys() const { return keys_; }
  const std::vector<Column>& deps() const { return deps_; }
  int32_t accOffset(int i) const { return accOffsets_[i]; }
  int32_t accNullOffset(int i) const { return accNullOffsets_[i]; }
  int32_t nextOffset() const { return nextOffset_; }
  int32_t rowIdOffset() const { return rowIdOffset_; }
  const std::vector<char*>& rows() const { return rows_; }
  int64_t numRows() const { return static_cast<int64_t>(rows_.size()); }

 private:
  std::vector<Column> keys_, deps_;
  std::vector<int32_t> accOffsets_, accNullOffsets_;
  int32_t nextOffset_ = -1;
  int32_t rowIdOffset_ = 0;
  int32_t rowSize_ = 0;
  std::vector<std::unique_ptr<char[]>> chunks_, strings_;
  int64_t chunkSize_ = 0, chunkUsed_ = 0;
  std::vector<char*> rows_;
};

// HashLookup (exec/HashTable.h:68-115).
struct HashLookup {
  std::vector<int32_t> rows;
  std::vector<uint64_t> hashes;
  std::vector<uint64_t> normalizedKeys;
  std::vector<char*> hits;
  std::vector<int32_t> newGroups;
  void reset(int32_t n) {
    rows.clear();
    hashes.resize(n);
    hits.resize(n);
    std::fill(hits.begin(), hits.end(), nullptr);
    newGroups.clear();
  }
};

class HashTable {
 public:
  static constexpr uint64_t kArrayHashMaxSize = 2L << 20;  // HashTable.h:146
  static constexpr double kLoadFactor = 0.7;               // HashTable.h:143
  static constexpr int kBucketSize = 128;                  // HashTable.h:897-930

  HashTable(const std::vector<int32_t>& keyKinds, const std::vector<int32_t>& accBytes,
            const std::vector<int32_t>& depKinds, bool isJoinBuild, bool allowDuplicates,
            bool ignoreNullKeys)
      : isJoinBuild_(isJoinBuild), allowDuplicates_(allowDuplicates),
        ignoreNullKeys_(ignoreNullKeys) {
    for (auto k : keyKinds) {
      hashers_.emplace_back(k);
      if (!hashers_.back().typeSupportsValueIds()) {
        hashMode_ = HashMode::kHash;  // HashTable.cpp:68-73
      }
    }
    rows_ = std::make_unique<RowContainer>(keyKinds, accBytes, depKinds,
                                           isJoinBuild && allowDuplicates);
  }
  ~HashTable() { std::free(table_); }

  HashMode hashMode() const { return hashMode_; }
  uint64_t capacity() const { return capacity_; }
  int64_t numDistinct() const { return numDistinct_; }
  int64_t numRehashes() const { return numRehashes_; }
  bool hasDuplicates() const { return hasDuplicates_; }
  RowContainer* rows() { return rows_.get(); }
  std::vector<VectorHasher>& hashers() { return hashers_; }

  // GroupingSet::createHashTable with hash_adaptivity_enabled = false
  // (GroupingSet.cpp:494-496) -> forceGenericHashMode.
  void forceGenericHashMode() {
    if (hashMode_ != HashMode::kHash) {
      setHashMode(HashMode::kHash, 0);
    }
  }

  // HashTable::prepareForGroupProbe (HashTable.cpp:2633-2677). 'rows' is a
  // mutable selection bitmap over [0, numRows).
  void prepareForGroupProbe(HashLookup& lookup, const std::vector<Decoded>& keys, int32_t numRows,
                            std::vector<uint64_t>& rows) {
    if (ignoreNullKeys_) {
      deselectRowsWithNulls(keys, numRows, rows);  // OperatorUtils.cpp:173-189
    }
    for (;;) {
      lookup.reset(numRows);
      bool rehash = false;
      for (size_t i = 0; i < hashers_.size(); ++i) {
        if (hashMode_ != HashMode::kHash) {
          if (!hashers_[i].computeValueIds(keys[i], numRows, rows.data(), lookup.hashes.data())) {
            rehash = true;
          }
        } else {
          hashers_[i].hash(keys[i], numRows, rows.data(), i > 0, lookup.hashes.data());
        }
      }
      if ((rehash || capacity_ == 0) && hashMode_ != HashMode::kHash) {
        decideHashMode(numRows);
        continue;
      }
      break;
    }
    for (int32_t r = 0; r < numRows; ++r) {  // populateLookupRows (:2613-2622)
      if (bitSet(rows.data(), r)) {
        lookup.rows.push_back(r);
      }
    }
  }

  // HashTable::groupProbe (HashTable.cpp:470-520).
  void groupProbe(HashLookup& lookup, const std::vector<Decoded>& keys) {
    if (lookup.rows.empty()) {
      return;
    }
    if (hashMode_ == HashMode::kArray) {
      arrayGroupProbe(lookup, keys);
      return;
    }
    checkSize(static_cast<int32_t>(lookup.rows.size()), false);
    if (hashMode_ == HashMode::kNormalizedKey) {
      populateNormalizedKeys(lookup);
    }
    // The reference walks 4 rows in lock step with prefetch (:487-513); the
    // visible behaviour is that of probing rows one by one in ascending order.
    constexpr int kAhead = 8;
    const size_t n = lookup.rows.size();
    for (size_t i = 0; i < n; ++i) {
      if (i + kAhead < n) {
        __builtin_prefetch(table_ + bucketOffset(lookup.hashes[lookup.rows[i + kAhead]]));
      }
      int32_t row = lookup.rows[i];
      lookup.hits[row] = fullProbeInsert(lookup, keys, row);
    }
  }

  // --- join side -----------------------------------------------------------

  // HashBuild::addInput row append (HashBuild.cpp:577-597).
  char* appendJoinRow(const std::vector<Decoded>& keys, const std::vector<Decoded>& deps,
                      int32_t index) {
    char* row = rows_->newRow();
    for (size_t i = 0; i < keys.size(); ++i) {
      rows_->store(keys[i], index, row, rows_->keys()[i]);
    }
    for (size_t i = 0; i < deps.size(); ++i) {
      rows_->store(deps[i], index, row, rows_->deps()[i]);
    }
    return row;
  }

  // Statistics pass over build keys (HashBuild.cpp:562-568: computeValueIds for
  // its analysis side effect only).
  void analyzeJoinKeys(const std::vector<Decoded>& keys, int32_t numRows, const uint64_t* rows) {
    if (hashMode_ == HashMode::kHash) {
      return;
    }
    std::vector<uint64_t> scratch(numRows);
    for (size_t i = 0; i < hashers_.size(); ++i) {
      hashers_[i].computeValueIds(keys[i], numRows, rows, scratch.data());
    }
  }

  // HashTable::prepareJoinTable (HashTable.cpp:1989-2069): merge the other
  // build drivers' hashers and rows, decide the mode, insert everything.
  void prepareJoinTable(std::vector<HashTable*> others) {
    otherTables_ = std::move(others);
    bool useValueIds = hashMode_ != HashMode::kHash;
    for (auto* other : otherTables_) {
      if (other->hashMode_ == HashMode::kHash) {
        useValueIds = false;
      }
    }
    if (useValueIds) {
      for (auto* other : otherTables_) {
        for (size_t i = 0; i < hashers_.size(); ++i) {
          hashers_[i].merge(other->hashers_[i], VectorHasher::kMaxDistinct);
        }
      }
      for (auto& h : hashers_) {
        uint64_t asRange, asDistincts;
        h.cardinality(0, asRange, asDistincts);
        if (asRange == VectorHasher::kRangeTooLarge &&
            asDistincts == VectorHasher::kRangeTooLarge) {
          useValueIds = false;
          break;
        }
      }
    }
    numDistinct_ = rows_->numRows();
    for (auto* other : otherTables_) {
      numDistinct_ += other->rows_->numRows();
    }
    if (!useValueIds) {
      if (hashMode_ != HashMode::kHash) {
        setHashMode(HashMode::kHash, 0);
      } else {
        checkSize(0, true);
      }
    } else {
      decideHashMode(0);
    }
  }

  // HashTable::joinProbe (HashTable.cpp:610-652). lookup.rows / hashes must be
  // prepared by prepareForJoinProbe.
  void prepareForJoinProbe(HashLookup& lookup, const std::vector<Decoded>& keys, int32_t numRows,
                           std::vector<uint64_t>& rows, bool nullAsValue = false) {
    if (!nullAsValue) {  // HashProbe.cpp:787-789
      deselectRowsWithNulls(keys, numRows, rows);
    }
    lookup.reset(numRows);
    for (size_t i = 0; i < hashers_.size(); ++i) {
      if (hashMode_ != HashMode::kHash) {
        hashers_[i].lookupValueIds(keys[i], numRows, rows.data(), lookup.hashes.data());
      } else {
        hashers_[i].hash(keys[i], numRows, rows.data(), i > 0, lookup.hashes.data());
      }
    }
    for (int32_t r = 0; r < numRows; ++r) {
      if (bitSet(rows.data(), r)) {
        lookup.rows.push_back(r);
      }
    }
  }

  void joinProbe(HashLookup& lookup, const std::vector<Decoded>& keys) {
    if (lookup.rows.empty() || numDistinct_ == 0) {
      return;
    }
    if (hashMode_ == HashMode::kArray) {  // arrayJoinProbe (:655-694)
      for (auto row : lookup.rows) {
        uint64_t index = lookup.hashes[row];
        lookup.hits[row] = index < capacity_ ? arrayTable()[index] : nullptr;
      }
      return;
    }
    if (hashMode_ == HashMode::kNormalizedKey) {
      populateNormalizedKeys(lookup);
    }
    constexpr int kAhead = 16;  // the reference keeps 64 probes in flight (:697-725)
    const size_t n = lookup.rows.size();
    for (size_t i = 0; i < n; ++i) {
      if (i + kAhead < n) {
        __builtin_prefetch(table_ + bucketOffset(lookup.hashes[lookup.rows[i + kAhead]]));
      }
      int32_t row = lookup.rows[i];
      lookup.hits[row] = probeOnly(lookup, keys, row);
    }
  }

  char* nextRow(char* row) const {  // duplicate chain, nextOffset_
    if (rows_->nextOffset() < 0) {
      return nullptr;
    }
    char* next;
    std::memcpy(&next, row + rows_->nextOffset(), 8);
    return next;
  }

 private:
  char** arrayTable() { return reinterpret_cast<char**>(table_); }

  static void deselectRowsWithNulls(const std::vector<Decoded>& keys, int32_t numRows,
                                    std::vector<uint64_t>& rows) {
    for (auto& d : keys) {
      if (!d.c->nulls) {
        continue;
      }
      for (int32_t r = 0; r < numRows; ++r) {
        if (bitSet(rows.data(), r) && d.isNull(r)) {
          setBit(rows.data(), r, false);
        }
      }
    }
  }

  int32_t reservePct() const { return (isJoinBuild_ && allowDuplicates_) ? 0 : 50; }  // HashTable.h:1213
  bool joinBuildNoDuplicates() const { return isJoinBuild_ && !allowDuplicates_; }

  static uint64_t safeMul(uint64_t a, uint64_t b) {  // HashTable.cpp:1665-1676
    constexpr uint64_t kMax = ~0ULL;
    if (a == kMax || b == kMax) {
      return kMax;
    }
    uint64_t r;
    if (__builtin_mul_overflow(a, b, &r)) {
      return kMax;
    }
    return r;
  }

  // All stored rows of this table and the merged ones, container 0 first
  // (rehash, HashTable.cpp:1569-1595).
  template <typename F>
  void forEachRow(F f) {
    for (char* r : rows_->rows()) {
      f(rows_.get(), r);
    }
    for (auto* other : otherTables_) {
      for (char* r : other->rows_->rows()) {
        f(other->rows_.get(), r);
      }
    }
  }

  // HashTable::analyze (HashTable.cpp:1631-1663).
  bool analyze() {
    for (size_t i = 0; i < hashers_.size(); ++i) {
      auto& hasher = hashers_[i];
      if (!hasher.isRange()) {
        continue;
      }
      uint64_t rangeSize, distinctSize;
      hasher.cardinality(0, rangeSize, distinctSize);
      if (distinctSize == VectorHasher::kRangeTooLarge &&
          rangeSize == VectorHasher::kRangeTooLarge) {
        return false;
      }
      const auto& col = rows_->keys()[i];
      for (char* row : rows_->rows()) {
        if (row[col.nullOffset]) {
          continue;
        }
        analyzeStored(hasher, row, col);
      }
    }
    return true;
  }

  static void analyzeStored(VectorHasher& hasher, const char* row, const RowContainer::Column& col) {
    if (col.kind == VX355_BOOLEAN) {
      return;
    }
    if (col.kind == VX355_VARCHAR || col.kind == VX355_VARBINARY) {
      StringView sv;
      std::memcpy(&sv, row + col.offset, 16);
      hasher.analyzeString(sv.data(), sv.size);
    } else if (col.kind == VX355_TIMESTAMP) {
      Timestamp ts;
      std::memcpy(&ts, row + col.offset, 16);
      hasher.analyzeInt(ts.seconds * 1000 + static_cast<int64_t>(ts.nanos / 1000000));
    } else {
      int64_t v;
      std::memcpy(&v, row + col.offset, 8);
      hasher.analyzeInt(v);
    }
  }

  static uint64_t valueIdStored(VectorHasher& hasher, const char* row,
                                const RowContainer::Column& col) {
    if (row[col.nullOffset]) {
      return 0;
    }
    if (col.kind == VX355_BOOLEAN) {
      int64_t v;
      std::memcpy(&v, row + col.offset, 8);
      return hasher.valueIdBool(v != 0);
    }
    if (col.kind == VX355_VARCHAR || col.kind == VX355_VARBINARY) {
      StringView sv;
      std::memcpy(&sv, row + col.offset, 16);
      return hasher.valueIdString(sv.data(), sv.size);
    }
    if (col.kind == VX355_TIMESTAMP) {
      Timestamp ts;
      std::memcpy(&ts, row + col.offset, 16);
      if (ts.nanos % 1000000 != 0) {
        return VectorHasher::kUnmappable;
      }
      return hasher.valueIdInt(ts.seconds * 1000 + static_cast<int64_t>(ts.nanos / 1000000));
    }
    int64_t v;
    std::memcpy(&v, row + col.offset, 8);
    return hasher.valueIdInt(v);
  }

  // HashTable::setHasherMode (HashTable.cpp:1727-1741).
  uint64_t setHasherMode(const std::vector<bool>& useRange) {
    uint64_t multiplier = 1;
    for (size_t i = 0; i < hashers_.size(); ++i) {
      multiplier = useRange[i] ? hashers_[i].enableValueRange(multiplier, reservePct())
                               : hashers_[i].enableValueIds(multiplier, reservePct());
      if (multiplier == VectorHasher::kRangeTooLarge) {
        throw std::runtime_error("setHasherMode: multiplier overflow");
      }
    }
    return multiplier;
  }

  void clearUseRange(std::vector<bool>& useRange) {  // :1743-1748
    for (size_t i = 0; i < hashers_.size(); ++i) {
      useRange[i] = hashers_[i].kind() == VX355_BOOLEAN;
    }
  }

  // HashTable::enableRangeWhereCan (HashTable.cpp:1682-1724).
  void enableRangeWhereCan(const std::vector<uint64_t>& rangeSizes,
                           const std::vector<uint64_t>& distinctSizes,
                           std::vector<bool>& useRange) {
    std::vector<size_t> indices(rangeSizes.size());
    std::vector<uint64_t> rangeMultipliers(rangeSizes.size(), ~0ULL);
    for (size_t i = 0; i < rangeSizes.size(); i++) {
      indices[i] = i;
      if (!useRange[i]) {
        rangeMultipliers[i] = rangeSizes[i] / distinctSizes[i];
      }
    }
    std::sort(indices.begin(), indices.end(),
              [&](auto i, auto j) { return rangeMultipliers[i] < rangeMultipliers[j]; });
    auto product = [&]() {
      uint64_t m = 1;
      for (size_t i = 0; i < rangeSizes.size(); ++i) {
        m = safeMul(m, useRange[i] ? rangeSizes[i] : distinctSizes[i]);
      }
      return m;
    };
    for (size_t i = 0; i < rangeSizes.size(); ++i) {
      if (!useRange[indices[i]]) {
        useRange[indices[i]] = true;
        if (product() == VectorHasher::kRangeTooLarge) {
          useRange[indices[i]] = false;
          return;
        }
      }
    }
  }

  // HashTable::decideHashMode (HashTable.cpp:1751-1839).
  void decideHashMode(int32_t numNew) {
    const size_t n = hashers_.size();
    std::vector<uint64_t> rangeSizes(n), distinctSizes(n);
    std::vector<bool> useRange(n);
    uint64_t bestWithReserve = 1, distinctsWithReserve = 1, rangesWithReserve = 1;
    if (numDistinct_ && (!isJoinBuild_ || joinBuildNoDuplicates())) {
      if (!analyze()) {
        setHashMode(HashMode::kHash, numNew);
        return;
      }
    }
    for (size_t i = 0; i < n; ++i) {
      hashers_[i].cardinality(reservePct(), rangeSizes[i], distinctSizes[i]);
      distinctsWithReserve = safeMul(distinctsWithReserve, distinctSizes[i]);
      rangesWithReserve = safeMul(rangesWithReserve, rangeSizes[i]);
      if (distinctSizes[i] == VectorHasher::kRangeTooLarge &&
          rangeSizes[i] != VectorHasher::kRangeTooLarge) {
        useRange[i] = true;
        bestWithReserve = safeMul(bestWithReserve, rangeSizes[i]);
      } else if (rangeSizes[i] != VectorHasher::kRangeTooLarge &&
                 rangeSizes[i] <= distinctSizes[i] * 20) {
        useRange[i] = true;
        bestWithReserve = safeMul(bestWithReserve, rangeSizes[i]);
      } else {
        bestWithReserve = safeMul(bestWithReserve, distinctSizes[i]);
      }
    }
    if (rangesWithReserve < kArrayHashMaxSize) {
      std::fill(useRange.begin(), useRange.end(), true);
      capacity_ = setHasherMode(useRange);
      setHashMode(HashMode::kArray, numNew);
      return;
    }
    if (bestWithReserve < kArrayHashMaxSize) {
      capacity_ = setHasherMode(useRange);
      setHashMode(HashMode::kArray, numNew);
      return;
    }
    if (rangesWithReserve != VectorHasher::kRangeTooLarge) {
      std::fill(useRange.begin(), useRange.end(), true);
      setHasherMode(useRange);
      setHashMode(HashMode::kNormalizedKey, numNew);
      return;
    }
    if (n == 1 && distinctsWithReserve > 10000) {
      setHashMode(HashMode::kHash, numNew);
      return;
    }
    if (distinctsWithReserve < kArrayHashMaxSize) {
      clearUseRange(useRange);
      capacity_ = setHasherMode(useRange);
      setHashMode(HashMode::kArray, numNew);
      return;
    }
    if (distinctsWithReserve == VectorHasher::kRangeTooLarge &&
        rangesWithReserve == VectorHasher::kRangeTooLarge) {
      setHashMode(HashMode::kHash, numNew);
      return;
    }
    if (bestWithReserve != VectorHasher::kRangeTooLarge) {
      enableRangeWhereCan(rangeSizes, distinctSizes, useRange);
    } else {
      clearUseRange(useRange);
    }
    setHasherMode(useRange);
    setHashMode(HashMode::kNormalizedKey, numNew);
  }

  // HashTable::setHashMode (HashTable.cpp:1599-1628).
  void setHashMode(HashMode mode, int32_t numNew) {
    if (mode == HashMode::kArray) {
      std::free(table_);
      table_ = static_cast<char*>(std::calloc(capacity_ ? capacity_ : 1, sizeof(char*)));
      hashMode_ = HashMode::kArray;
      rehash(true);
    } else if (mode == HashMode::kHash) {
      hashMode_ = HashMode::kHash;
      for (auto& h : hashers_) {
        h.resetStats();
      }
      capacity_ = 0;
      checkSize(numNew, true);
    } else {
      hashMode_ = HashMode::kNormalizedKey;
      capacity_ = 0;
      checkSize(numNew, true);
    }
  }

  static uint64_t nextPowerOfTwo(uint64_t size) {  // BitUtil.h:752-763
    if (size == 0) {
      return 0;
    }
    uint32_t bits = 63 - __builtin_clzll(size);
    uint64_t lower = 1ULL << bits;
    if (lower == size) {
      return size;
    }
    return 2 * lower;
  }

  // newHashTableEntries (HashTable.h:946-956).
  static uint64_t newHashTableEntries(uint64_t numDistincts, uint64_t numNew) {
    auto numNewEntries = std::max<uint64_t>(2048, nextPowerOfTwo(numNew * 2 + numDistincts));
    if (numDistincts + numNew > static_cast<uint64_t>(numNewEntries * kLoadFactor)) {
      numNewEntries *= 2;
    }
    return numNewEntries;
  }

  void allocateTables(uint64_t size) {  // HashTable.cpp:728-750
    capacity_ = size;
    const uint64_t byteSize = capacity_ * 8;
    sizeMask_ = byteSize - 1;
    numBuckets_ = byteSize / kBucketSize;
    bucketOffsetMask_ = sizeMask_ & ~static_cast<uint64_t>(kBucketSize - 1);
    std::free(table_);
    table_ = static_cast<char*>(std::aligned_alloc(kBucketSize, byteSize));
    std::memset(table_, 0, byteSize);
  }

  // HashTable::checkSize (HashTable.cpp:772-806).
  void checkSize(int32_t numNew, bool initNormalizedKeys) {
    const int64_t newNumDistincts = numNew + numDistinct_;
    if (table_ == nullptr || capacity_ == 0) {
      allocateTables(newHashTableEntries(numDistinct_, numNew));
      if (numDistinct_ > 0) {
        rehash(initNormalizedKeys);
      }
    } else if (newNumDistincts > static_cast<int64_t>(capacity_ * kLoadFactor)) {
      allocateTables(nextPowerOfTwo(std::max<int64_t>(newNumDistincts, capacity_) + 1));
      rehash(initNormalizedKeys);
    }
  }

  int64_t bucketOffset(uint64_t hash) const { return hash & bucketOffsetMask_; }
  int64_t nextBucketOffset(int64_t off) const { return sizeMask_ & (off + kBucketSize); }
  static uint8_t hashTag(uint64_t hash) { return static_cast<uint8_t>(hash >> 38) | 0x80; }

  // Bucket (HashTable.h:897-930): 16 tag bytes then 16 six-byte pointers.
  char* pointerAt(int64_t bucketOff, int slot) const {
    uint64_t p = 0;
    std::memcpy(&p, table_ + bucketOff + 16 + 6 * slot, 6);
    return reinterpret_cast<char*>(p);
  }
  void setSlot(int64_t bucketOff, int slot, uint8_t tag, char* row) {
    table_[bucketOff + slot] = static_cast<char>(tag);
    uint64_t p = reinterpret_cast<uint64_t>(row);
    std::memcpy(table_ + bucketOff + 16 + 6 * slot, &p, 6);
  }

  static uint16_t matchTags(const char* tags, uint8_t wanted) {
    uint16_t m = 0;
    for (int i = 0; i < 16; ++i) {
      m |= static_cast<uint16_t>(static_cast<uint8_t>(tags[i]) == wanted) << i;
    }
    return m;
  }

  // populateNormalizedKeys (HashTable.cpp:446-466).
  void populateNormalizedKeys(HashLookup& lookup) {
    lookup.normalizedKeys.resize(lookup.hashes.size());
    for (auto row : lookup.rows) {
      auto key = lookup.hashes[row];
      lookup.normalizedKeys[row] = key;
      lookup.hashes[row] = twangMix64(key);  // mixNormalizedKey (:442-444)
    }
  }

  bool compareKeys(const char* group, const std::vector<Decoded>& keys, int32_t row) const {
    // HashTable::compareKeys (HashTable.cpp:359-380)
    for (size_t i = 0; i < keys.size(); ++i) {
      if (!RowContainer::equals(group, rows_->keys()[i], keys[i], row)) {
        return false;
      }
    }
    return true;
  }

  // HashTable::insertEntry (HashTable.cpp:337-356).
  char* insertEntry(HashLookup& lookup, const std::vector<Decoded>& keys, int32_t row) {
    char* group = rows_->newRow();
    for (size_t i = 0; i < keys.size(); ++i) {
      rows_->store(keys[i], row, group, rows_->keys()[i]);
    }
    if (hashMode_ == HashMode::kNormalizedKey) {
      RowContainer::normalizedKey(group) = lookup.normalizedKeys[row];
    }
    ++numDistinct_;
    lookup.newGroups.push_back(row);
    return group;
  }

  // arrayGroupProbe (HashTable.cpp:560-607).
  void arrayGroupProbe(HashLookup& lookup, const std::vector<Decoded>& keys) {
    for (auto row : lookup.rows) {
      uint64_t index = lookup.hashes[row];
      char* group = arrayTable()[index];
      if (!group) {
        group = insertEntry(lookup, keys, row);
        arrayTable()[index] = group;
      }
      lookup.hits[row] = group;
    }
  }

  // ProbeState::fullProbe<kInsert> (HashTable.cpp:138-231) for group by.
  char* fullProbeInsert(HashLookup& lookup, const std::vector<Decoded>& keys, int32_t row) {
    const uint64_t hash = lookup.hashes[row];
    const uint8_t tag = hashTag(hash);
    int64_t off = bucketOffset(hash);
    for (uint64_t probed = 0; probed < numBuckets_; ++probed) {
      const char* tags = table_ + off;
      uint16_t hits = matchTags(tags, tag);
      while (hits) {
        int slot = __builtin_ctz(hits);
        hits &= hits - 1;
        char* group = pointerAt(off, slot);
        bool same = hashMode_ == HashMode::kNormalizedKey
            ? RowContainer::normalizedKey(group) == lookup.normalizedKeys[row]
            : compareKeys(group, keys, row);
        if (same) {
          return group;
        }
      }
      uint16_t empty = matchTags(tags, 0);
      if (empty) {
        int slot = __builtin_ctz(empty);
        char* group = insertEntry(lookup, keys, row);
        setSlot(off, slot, tag, group);
        return group;
      }
      off = nextBucketOffset(off);
    }
    throw std::runtime_error("Have looped through all the buckets in table");
  }

  // ProbeState::fullProbe<kProbe> / joinNormalizedKeyFullProbe (:234-265).
  char* probeOnly(HashLookup& lookup, const std::vector<Decoded>& keys, int32_t row) {
    const uint64_t hash = lookup.hashes[row];
    const uint8_t tag = hashTag(hash);
    int64_t off = bucketOffset(hash);
    for (uint64_t probed = 0; probed < numBuckets_; ++probed) {
      const char* tags = table_ + off;
      uint16_t hits = matchTags(tags, tag);
      while (hits) {
        int slot = __builtin_ctz(hits);
        hits &= hits - 1;
        char* group = pointerAt(off, slot);
        bool same = hashMode_ == HashMode::kNormalizedKey
            ? RowContainer::normalizedKey(group) == lookup.normalizedKeys[row]
            : compareKeys(group, keys, row);
        if (same) {
          return group;
        }
      }
      if (matchTags(tags, 0)) {
        return nullptr;
      }
      off = nextBucketOffset(off);
    }
    return nullptr;
  }

  // Hash (or value id) of a stored row for rehash: HashTable::hashRows
  // (HashTable.cpp:809-857). Returns false if a key became unmappable.
  bool hashStoredRow(RowContainer* container, char* row, bool initNormalizedKeys, uint64_t& hash) {
    if (!initNormalizedKeys && hashMode_ == HashMode::kNormalizedKey) {
      hash = twangMix64(RowContainer::normalizedKey(row));
      return true;
    }
    hash = 0;
    for (size_t i = 0; i < hashers_.size(); ++i) {
      const auto& col = container->keys()[i];
      if (hashMode_ == HashMode::kHash) {
        uint64_t h = RowContainer::hashStored(row, col);
        hash = i == 0 ? h : hashMix(hash, h);
      } else {
        uint64_t id = valueIdStored(hashers_[i], row, col);
        if (id == VectorHasher::kUnmappable) {
          return false;
        }
        hash = hashers_[i].multiplier() == 1 ? id : hash + hashers_[i].multiplier() * id;
      }
    }
    if (hashMode_ == HashMode::kNormalizedKey && initNormalizedKeys) {
      RowContainer::normalizedKey(row) = hash;
      hash = twangMix64(hash);
    }
    return true;
  }

  static bool storedKeysEqual(RowContainer* ca, const char* a, RowContainer* cb, const char* b) {
    for (size_t i = 0; i < ca->keys().size(); ++i) {
      const auto& x = ca->keys()[i];
      const auto& y = cb->keys()[i];
      bool an = a[x.nullOffset], bn = b[y.nullOffset];
      if (an || bn) {
        if (an != bn) {
          return false;
        }
        continue;
      }
      if (x.kind == VX355_VARCHAR || x.kind == VX355_VARBINARY) {
        StringView s, t;
        std::memcpy(&s, a + x.offset, 16);
        std::memcpy(&t, b + y.offset, 16);
        if (s.size != t.size || std::memcmp(s.data(), t.data(), s.size) != 0) {
          return false;
        }
      } else if (x.kind == VX355_DOUBLE) {
        double s, t;
        std::memcpy(&s, a + x.offset, 8);
        std::memcpy(&t, b + y.offset, 8);
        if (!((std::isnan(s) && std::isnan(t)) || s == t)) {
          return false;
        }
      } else if (x.kind == VX355_REAL) {
        float s, t;
        std::memcpy(&s, a + x.offset, 4);
        std::memcpy(&t, b + y.offset, 4);
        if (!((std::isnan(s) && std::isnan(t)) || s == t)) {
          return false;
        }
      } else if (std::memcmp(a + x.offset, b + y.offset, x.width) != 0) {
        return false;
      }
    }
    return true;
  }

  // pushNext (HashTable.cpp:1412-1418): chain = head, then later rows in
  // reverse insertion order.
  void pushNext(char* head, char* row) {
    hasDuplicates_ = true;
    int32_t no = rows_->nextOffset();
    char* headNext;
    std::memcpy(&headNext, head + no, 8);
    std::memcpy(row + no, &headNext, 8);
    std::memcpy(head + no, &row, 8);
  }

  // HashTable::rehash (HashTable.cpp:1541-1596) -> insertBatch (:1310) ->
  // insertForGroupBy (:1327-1391) / insertForJoin (:1518-1538) ->
  // buildFullProbe (:1422-1478) / arrayPushRow (:1394-1409).
  void rehash(bool initNormalizedKeys) {
    ++numRehashes_;
    if (canApplyParallelJoinBuild()) {
      if (parallelJoinBuild(initNormalizedKeys)) {
        return;
      }
      // (a key became unmappable: the serial path below finds it again and falls back to kHash)
    }
    bool failed = false;
    int64_t distinct = 0;
    forEachRow([&](RowContainer* container, char* row) {
      if (failed) {
        return;
      }
      uint64_t hash;
      bool init = initNormalizedKeys || container != rows_.get();
      if (!hashStoredRow(container, row, init, hash)) {
        failed = true;
        return;
      }
      if (hashMode_ == HashMode::kArray) {
        if (hash >= capacity_) {
          throw std::runtime_error("array index out of range in rehash");
        }
        char*& slot = arrayTable()[hash];
        if (isJoinBuild_) {
          if (slot == nullptr) {
            slot = row;
            ++distinct;
          } else if (allowDuplicates_) {
            // arrayPushRow: new.next = table[i]; table[i] = new.
            hasDuplicates_ = true;
            std::memcpy(row + rows_->nextOffset(), &slot, 8);
            slot = row;
          }
        } else {
          slot = row;
          ++distinct;
        }
        return;
      }
      const uint8_t tag = hashTag(hash);
      int64_t off = bucketOffset(hash);
      for (uint64_t probed = 0; probed < numBuckets_; ++probed) {
        const char* tags = table_ + off;
        if (isJoinBuild_) {
          uint16_t hits = matchTags(tags, tag);
          bool done = false;
          while (hits) {
            int slot = __builtin_ctz(hits);
            hits &= hits - 1;
            char* head = pointerAt(off, slot);
            bool same = hashMode_ == HashMode::kNormalizedKey
                ? RowContainer::normalizedKey(head) == RowContainer::normalizedKey(row)
                : storedKeysEqual(rows_.get(), head, container, row);
            if (same) {
              if (allowDuplicates_) {
                pushNext(head, row);
              }
              done = true;
              break;
            }
          }
          if (done) {
            return;
          }
        }
        uint16_t empty = matchTags(tags, 0);
        if (empty) {
          setSlot(off, __builtin_ctz(empty), tag, row);
          ++distinct;
          return;
        }
        off = nextBucketOffset(off);
      }
      throw std::runtime_error("Have looped through all the buckets in table");
    });
    if (failed) {
      setHashMode(HashMode::kHash, 0);
      return;
    }
    if (isJoinBuild_) {
      numDistinctKeys_ = distinct;
    }
  }

  // HashTable::canApplyParallelJoinBuild (HashTable.cpp:984-1000): a join build with peer tables whose
  // share of the table is large enough; off unless the caller asked for build threads
  // (setBuildThreads: the parity tests run the serial path, bench.py's multi-thread CPU leg this one).
  bool canApplyParallelJoinBuild() const {
    if (!isJoinBuild_ || buildThreads_ <= 1 || hashMode_ == HashMode::kArray || otherTables_.empty() ||
        otherTables_.size() > 254) {
      return false;
    }
    if (hashMode_ == HashMode::kNormalizedKey) {
      // (this restatement's distinct-value mode looks ids up through a method that can also insert:
      // only range-mode hashers - a subtraction - are read from several threads)
      for (const auto& h : hashers_) {
        if (!h.isRange()) {
          return false;
        }
      }
    }
    return (capacity_ / (1 + otherTables_.size())) > kMinTableSizeForParallelJoinBuild;
  }

  // HashTable::parallelJoinBuild (HashTable.cpp:1003-1203): one partition per build table, in terms
  // of ranges of bucket offsets. Step 1 (partitionRows, :1231-1263): every table's rows are hashed
  // and assigned the partition of their first bucket, one thread per table. Step 2
  // (buildJoinPartition, :1265-1308): one thread per partition inserts the rows of ALL tables that
  // start in its range, probing no further than the range's end; rows that would run past it are
  // collected as overflow. Step 3: the overflow rows are inserted serially. false = a key was
  // unmappable (nothing inserted yet: the caller takes the serial path, which handles it).
  bool parallelJoinBuild(bool initNormalizedKeys) {
    const int numPartitions = 1 + static_cast<int>(otherTables_.size());
    std::vector<RowContainer*> containers = {rows_.get()};
    for (auto* other : otherTables_) {
      containers.push_back(other->rows_.get());
    }
    std::vector<uint64_t> bounds(numPartitions + 1);
    for (int i = 0; i < numPartitions; ++i) {
      bounds[i] = (((sizeMask_ + 1) / numPartitions) * i + kBucketSize - 1) / kBucketSize * kBucketSize;
    }
    bounds[numPartitions] = sizeMask_ + 1;
    std::vector<std::vector<uint64_t>> hashes(numPartitions);
    std::vector<std::vector<uint8_t>> partOf(numPartitions);
    std::vector<char> bad(numPartitions, 0);
    auto inThreads = [&](auto&& work) {
      // (the reference runs the last step on the calling thread; buildThreads_ caps the workers)
      std::vector<std::thread> threads;
      std::atomic<int> next{0};
      const int workers = std::min(buildThreads_, numPartitions);
      for (int t = 0; t < workers; ++t) {
        threads.emplace_back([&] {
          for (int i = next.fetch_add(1); i < numPartitions; i = next.fetch_add(1)) {
            work(i);
          }
        });
      }
      for (auto& t : threads) {
        t.join();
      }
    };
    inThreads([&](int c) {
      RowContainer* container = containers[c];
      const auto& rows = container->rows();
      hashes[c].resize(rows.size());
      partOf[c].resize(rows.size());
      const bool init = initNormalizedKeys || container != rows_.get();
      for (size_t r = 0; r < rows.size(); ++r) {
        uint64_t hash;
        if (!hashStoredRow(container, rows[r], init, hash)) {
          bad[c] = 1;
          return;
        }
        hashes[c][r] = hash;
        const uint64_t off = static_cast<uint64_t>(bucketOffset(hash));
        int part = static_cast<int>(std::upper_bound(bounds.begin(), bounds.end(), off) - bounds.begin()) - 1;
        partOf[c][r] = static_cast<uint8_t>(part);
      }
    });
    for (char b : bad) {
      if (b) {
        return false;
      }
    }
    std::vector<std::vector<std::pair<char*, uint64_t>>> overflow(numPartitions);
    std::vector<std::vector<RowContainer*>> overflowContainer(numPartitions);
    std::vector<int64_t> distinct(numPartitions, 0);
    std::vector<char> duplicates(numPartitions, 0);
    inThreads([&](int p) {
      const int64_t end = static_cast<int64_t>(bounds[p + 1]);
      for (int c = 0; c < numPartitions; ++c) {
        const auto& rows = containers[c]->rows();
        for (size_t r = 0; r < rows.size(); ++r) {
          if (partOf[c][r] != p) {
            continue;
          }
          bool dup = false;
          const int placed = insertStoredRow(containers[c], rows[r], hashes[c][r], end, &dup);
          if (placed < 0) {
            overflow[p].emplace_back(rows[r], hashes[c][r]);
            overflowContainer[p].push_back(containers[c]);
          } else {
            distinct[p] += placed;
            duplicates[p] = duplicates[p] || dup;
          }
        }
      }
    });
    int64_t total = 0;
    for (int p = 0; p < numPartitions; ++p) {
      total += distinct[p];
      hasDuplicates_ = hasDuplicates_ || duplicates[p];
    }
    for (int p = 0; p < numPartitions; ++p) {
      for (size_t i = 0; i < overflow[p].size(); ++i) {
        bool dup = false;
        const int placed = insertStoredRow(overflowContainer[p][i], overflow[p][i].first, overflow[p][i].second, -1, &dup);
        total += placed > 0 ? 1 : 0;
        hasDuplicates_ = hasDuplicates_ || dup;
      }
    }
    numDistinctKeys_ = total;
    return true;
  }

  // buildFullProbe for a stored row (HashTable.cpp:1422-1478): 1 = a new key took a slot, 0 = the row
  // joined (or, without duplicates, was dropped at) an existing key, -1 = the probe sequence reached
  // 'end' (the partition's last bucket offset + 1; -1 = no limit, wrap around) without a free slot.
  int insertStoredRow(RowContainer* container, char* row, uint64_t hash, int64_t end, bool* duplicate) {
    const uint8_t tag = hashTag(hash);
    int64_t off = bucketOffset(hash);
    for (uint64_t probed = 0; probed < numBuckets_; ++probed) {
      const char* tags = table_ + off;
      uint16_t hits = matchTags(tags, tag);
      while (hits) {
        int slot = __builtin_ctz(hits);
        hits &= hits - 1;
        char* head = pointerAt(off, slot);
        bool same = hashMode_ == HashMode::kNormalizedKey
            ? RowContainer::normalizedKey(head) == RowContainer::normalizedKey(row)
            : storedKeysEqual(rows_.get(), head, container, row);
        if (same) {
          if (allowDuplicates_) {
            // pushNext without touching the shared flag (the caller merges 'duplicate')
            int32_t no = rows_->nextOffset();
            char* headNext;
            std::memcpy(&headNext, head + no, 8);
            std::memcpy(row + no, &headNext, 8);
            std::memcpy(head + no, &row, 8);
            *duplicate = true;
          }
          return 0;
        }
      }
      uint16_t empty = matchTags(tags, 0);
      if (empty) {
        setSlot(off, __builtin_ctz(empty), tag, row);
        return 1;
      }
      if (end >= 0) {
        off += kBucketSize;
        if (off >= end) {
          return -1;
        }
      } else {
        off = nextBucketOffset(off);
      }
    }
    throw std::runtime_error("Have looped through all the buckets in table");
  }

 public:
  int64_t numDistinctKeys() const { return isJoinBuild_ ? numDistinctKeys_ : numDistinct_; }
  /// Threads of the parallel join build (1 = serial, the default).
  void setBuildThreads(int n) { buildThreads_ = n < 1 ? 1 : n; }
  static constexpr uint64_t kMinTableSizeForParallelJoinBuild = 1000;  // QueryConfig::minTableRowsForParallelJoinBuild

 private:
  std::vector<VectorHasher> hashers_;
  std::unique_ptr<RowContainer> rows_;
  std::vector<HashTable*> otherTables_;
  HashMode hashMode_ = HashMode::kArray;
  bool isJoinBuild_, allowDuplicates_, ignoreNullKeys_;
  bool hasDuplicates_ = false;
  char* table_ = nullptr;
  uint64_t capacity_ = 0, sizeMask_ = 0, numBuckets_ = 0, bucketOffsetMask_ = 0;
  int64_t numDistinct_ = 0, numDistinctKeys_ = 0, numRehashes_ = 0;
  int buildThreads_ = 1;
};

}  // namespace orc
