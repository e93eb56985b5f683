// TEST INFRASTRUCTURE ONLY — CPU oracle, see oracle.h.
// Restatement of exec/VectorHasher.{h,cpp,-inl.h}: hashing, value ids, range /
// distinct statistics and the mode switches.
#pragma once
#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#include "hashing.h"

namespace orc {

class VectorHasher {
 public:
  static constexpr uint64_t kUnmappable = ~0ULL;                // VectorHasher.h:131
  static constexpr int64_t kMaxRange = ~0ULL >> 5;              // :135
  static constexpr uint64_t kRangeTooLarge = ~0ULL;             // :136
  static constexpr int32_t kMaxDistinct = 100000;               // :138
  static constexpr int32_t kNoLimit = -1;                       // :142
  static constexpr uint32_t kStringASRangeMaxSize = 7;          // :379

  explicit VectorHasher(int32_t kind) : kind_(kind) {}

  int32_t kind() const { return kind_; }
  bool isRange() const { return isRange_; }
  uint64_t multiplier() const { return multiplier_; }
  int64_t min() const { return min_; }
  int64_t max() const { return max_; }

  // VectorHasher.h:338-357.
  bool typeSupportsValueIds() const {
    switch (kind_) {
      case VX355_BOOLEAN:
      case VX355_TINYINT:
      case VX355_SMALLINT:
      case VX355_INTEGER:
      case VX355_BIGINT:
      case VX355_VARCHAR:
      case VX355_VARBINARY:
      case VX355_TIMESTAMP:
        return true;
      default:
        return false;
    }
  }

  // VectorHasher::hash (VectorHasher.cpp:567-584) -> hashValues (:86-126).
  // 'rows' may be null (= all rows).
  void hash(const Decoded& d, int32_t numRows, const uint64_t* rows, bool mix, uint64_t* result) {
    if (d.isConstant()) {
      uint64_t h = d.isNull(0) ? kNullHash : d.hashAt(0);
      for (int32_t r = 0; r < numRows; ++r) {
        if (rows && !bitSet(rows, r)) {
          continue;
        }
        result[r] = mix ? hashMix(result[r], h) : h;
      }
      return;
    }
    for (int32_t r = 0; r < numRows; ++r) {
      if (rows && !bitSet(rows, r)) {
        continue;
      }
      uint64_t h = d.isNull(r) ? kNullHash : d.hashAt(r);
      result[r] = mix ? hashMix(result[r], h) : h;
    }
  }

  // VectorHasher::computeValueIds (VectorHasher.cpp:354-360) -> makeValueIds
  // (:128-161) -> makeValueIdForOneRow (:196-224). Returns false when a value
  // was unmappable; keeps analysing the remaining rows so that the caller's
  // decideHashMode sees complete statistics.
  bool computeValueIds(const Decoded& d, int32_t numRows, const uint64_t* rows, uint64_t* result) {
    if (d.isConstant()) {
      int32_t first = firstSelected(numRows, rows);
      if (first < 0) {
        return true;
      }
      uint64_t id = d.isNull(first) ? 0 : valueIdAt(d, first);
      if (id == kUnmappable) {
        analyzeAt(d, first);
        return false;
      }
      for (int32_t r = 0; r < numRows; ++r) {
        if (rows && !bitSet(rows, r)) {
          continue;
        }
        result[r] = multiplier_ == 1 ? id : result[r] + multiplier_ * id;
      }
      return true;
    }
    bool success = true;
    for (int32_t r = 0; r < numRows; ++r) {
      if (rows && !bitSet(rows, r)) {
        continue;
      }
      if (d.isNull(r)) {
        if (multiplier_ == 1) {
          result[r] = 0;
        }
        continue;
      }
      if (!success) {
        analyzeAt(d, r);
        continue;
      }
      uint64_t id = valueIdAt(d, r);
      if (id == kUnmappable) {
        success = false;
        analyzeAt(d, r);
      } else {
        result[r] = multiplier_ == 1 ? id : result[r] + multiplier_ * id;
      }
    }
    return success;
  }

  // VectorHasher::lookupValueIds (VectorHasher.cpp:550-565) ->
  // lookupValueIdsTyped (:408-492): unmappable rows are deselected.
  void lookupValueIds(const Decoded& d, int32_t numRows, uint64_t* rows, uint64_t* result) const {
    if (d.isConstant()) {
      int32_t first = firstSelected(numRows, rows);
      if (first < 0) {
        return;
      }
      if (d.isNull(first)) {
        if (multiplier_ == 1) {
          for (int32_t r = 0; r < numRows; ++r) {
            if (bitSet(rows, r)) {
              result[r] = 0;
            }
          }
        }
        return;
      }
      uint64_t id = lookupValueIdAt(d, first);
      for (int32_t r = 0; r < numRows; ++r) {
        if (!bitSet(rows, r)) {
          continue;
        }
        if (id == kUnmappable) {
          setBit(rows, r, false);
        } else {
          result[r] = multiplier_ == 1 ? id : result[r] + multiplier_ * id;
        }
      }
      return;
    }
    for (int32_t r = 0; r < numRows; ++r) {
      if (!bitSet(rows, r)) {
        continue;
      }
      if (d.isNull(r)) {
        if (multiplier_ == 1) {
          result[r] = 0;
        }
        continue;
      }
      uint64_t id = lookupValueIdAt(d, r);
      if (id == kUnmappable) {
        setBit(rows, r, false);
        continue;
      }
      result[r] = multiplier_ == 1 ? id : result[r] + multiplier_ * id;
    }
  }

  // VectorHasher::cardinality (VectorHasher.cpp:853-904).
  void cardinality(int32_t reservePct, uint64_t& asRange, uint64_t& asDistincts) {
    if (!typeSupportsValueIds()) {
      asRange = kRangeTooLarge;
      asDistincts = kRangeTooLarge;
      return;
    }
    if (kind_ == VX355_BOOLEAN) {
      hasRange_ = true;
      asRange = 3;
      asDistincts = 3;
      return;
    }
    int64_t signedRange;
    if (!hasRange_ || rangeOverflow_) {
      asRange = kRangeTooLarge;
    } else if (__builtin_sub_overflow(max_, min_, &signedRange)) {
      setRangeOverflow();
      asRange = kRangeTooLarge;
    } else if (signedRange < kMaxRange) {
      int64_t mn = min_;
      int64_t mx = max_;
      extendRange(kind_, reservePct, mn, mx);
      asRange = (mx - mn) + 2;
    } else {
      setRangeOverflow();
      asRange = kRangeTooLarge;
    }
    if (distinctOverflow_) {
      asDistincts = kRangeTooLarge;
      return;
    }
    asDistincts = addIdReserve(numDistinct(), reservePct) + 1;
  }

  // VectorHasher::enableValueIds (VectorHasher.cpp:906-921).
  uint64_t enableValueIds(uint64_t multiplier, int32_t reservePct) {
    multiplier_ = multiplier;
    rangeSize_ = addIdReserve(numDistinct(), reservePct) + 1;
    isRange_ = false;
    uint64_t result;
    if (__builtin_mul_overflow(multiplier_, rangeSize_, &result)) {
      return kRangeTooLarge;
    }
    return result;
  }

  // VectorHasher::enableValueRange (VectorHasher.cpp:923-944).
  uint64_t enableValueRange(uint64_t multiplier, int32_t reservePct) {
    multiplier_ = multiplier;
    extendRange(kind_, reservePct, min_, max_);
    isRange_ = true;
    if (kind_ == VX355_BOOLEAN) {
      rangeSize_ = 3;
    } else {
      rangeSize_ = (max_ - min_) + 2;
    }
    uint64_t result;
    if (__builtin_mul_overflow(multiplier_, rangeSize_, &result)) {
      return kRangeTooLarge;
    }
    return result;
  }

  bool empty() const { return !hasRange_ && numDistinct() == 0; }  // VectorHasher.h:366-370

  // VectorHasher::merge (VectorHasher.cpp:958-1009).
  void merge(const VectorHasher& other, size_t maxNumDistinct) {
    if (kind_ == VX355_BOOLEAN) {
      return;
    }
    if (other.empty()) {
      return;
    }
    if (empty()) {
      hasRange_ = other.hasRange_;
      rangeOverflow_ = other.rangeOverflow_;
      distinctOverflow_ = other.distinctOverflow_;
      min_ = other.min_;
      max_ = other.max_;
      intIds_ = other.intIds_;
      strIds_ = other.strIds_;
      return;
    }
    if (hasRange_ && other.hasRange_ && !rangeOverflow_ && !other.rangeOverflow_) {
      min_ = std::min(min_, other.min_);
      max_ = std::max(max_, other.max_);
    } else {
      setRangeOverflow();
    }
    if (distinctOverflow_) {
      return;
    }
    if (other.distinctOverflow_) {
      setDistinctOverflow();
      return;
    }
    // Insertion order of 'other' is its id order.
    std::vector<std::pair<uint64_t, int64_t>> ints;
    for (auto& kv : other.intIds_) {
      ints.emplace_back(kv.second, kv.first);
    }
    std::sort(ints.begin(), ints.end());
    for (auto& p : ints) {
      if (intIds_.emplace(p.second, numDistinct() + 1).second && numDistinct() > maxNumDistinct) {
        setDistinctOverflow();
        return;
      }
    }
    std::vector<std::pair<uint64_t, std::string>> strs;
    for (auto& kv : other.strIds_) {
      strs.emplace_back(kv.second, kv.first);
    }
    std::sort(strs.begin(), strs.end());
    for (auto& p : strs) {
      if (strIds_.emplace(p.second, numDistinct() + 1).second && numDistinct() > maxNumDistinct) {
        setDistinctOverflow();
        return;
      }
    }
  }

  // VectorHasher::resetStats (VectorHasher.h): forget values seen.
  void resetStats() {
    intIds_.clear();
    strIds_.clear();
  }

  // Analyse one stored key value (HashTable::analyze -> VectorHasher::analyze,
  // VectorHasher.cpp:614-623).
  void analyzeInt(int64_t v) { analyzeInt64(v); }
  void analyzeString(const char* data, uint32_t size) { analyzeStr(data, size); }

  // Value ids of stored keys when a table is rehashed
  // (VectorHasher::computeValueIdsForRows, VectorHasher.cpp:362-381).
  uint64_t valueIdInt(int64_t v) { return valueIdInt64(v); }
  uint64_t valueIdString(const char* data, uint32_t size) { return valueIdStr(data, size); }
  uint64_t valueIdBool(bool b) const { return b ? 2 : 1; }

  size_t numDistinct() const { return intIds_.size() + strIds_.size(); }
  bool hasRange() const { return hasRange_; }
  bool rangeOverflow() const { return rangeOverflow_; }
  bool distinctOverflow() const { return distinctOverflow_; }
  uint64_t rangeSize() const { return rangeSize_; }

  // VectorHasher.h:383-387.
  static int64_t stringAsNumber(const char* data, int32_t size) {
    int64_t word = loadPartialWord(reinterpret_cast<const uint8_t*>(data), size);
    return size == 0 ? word : word + (1L << (size * 8));
  }

 private:
  static int32_t firstSelected(int32_t numRows, const uint64_t* rows) {
    for (int32_t r = 0; r < numRows; ++r) {
      if (!rows || bitSet(rows, r)) {
        return r;
      }
    }
    return -1;
  }

  // extendRange (VectorHasher.cpp:786-835).
  template <typename T>
  static void extendRangeT(int64_t reserve, int64_t& mn, int64_t& mx) {
    int64_t kMin = std::numeric_limits<T>::min();
    int64_t kMax = std::numeric_limits<T>::max();
    if (kMin + reserve + 1 > mn) {
      mn = kMin;
    } else {
      mn -= reserve;
    }
    if (kMax - reserve < mx) {
      mx = kMax;
    } else {
      mx += reserve;
    }
  }
  static void extendRange(int32_t kind, int32_t reservePct, int64_t& mn, int64_t& mx) {
    int64_t reserve = reservePct == 0 ? 0 : 2 + (mx - mn) * (reservePct / 100.0);
    switch (kind) {
      case VX355_BOOLEAN:
        break;
      case VX355_TINYINT:
        extendRangeT<int8_t>(reserve, mn, mx);
        break;
      case VX355_SMALLINT:
        extendRangeT<int16_t>(reserve, mn, mx);
        break;
      case VX355_INTEGER:
        extendRangeT<int32_t>(reserve, mn, mx);
        break;
      default:
        extendRangeT<int64_t>(reserve, mn, mx);
        break;
    }
  }
  // addIdReserve (VectorHasher.cpp:837-851).
  static int64_t addIdReserve(size_t numDistinct, int32_t reservePct) {
    if (numDistinct > static_cast<size_t>(kMaxDistinct)) {
      return numDistinct;
    }
    if (reservePct == kNoLimit) {
      return kMaxDistinct;
    }
    return std::min<int64_t>(kMaxDistinct, numDistinct * (1 + (reservePct / 100.0)));
  }

  void updateRange(int64_t v) {  // VectorHasher.h:603-614
    if (hasRange_) {
      if (v < min_) {
        min_ = v;
      } else if (v > max_) {
        max_ = v;
      }
    } else {
      hasRange_ = true;
      min_ = max_ = v;
    }
  }
  void setDistinctOverflow() {
    distinctOverflow_ = true;
    intIds_.clear();
    strIds_.clear();
  }
  void setRangeOverflow() {
    rangeOverflow_ = true;
    hasRange_ = false;
  }

  // valueId<T> (VectorHasher.h:560-580).
  uint64_t valueIdInt64(int64_t v) {
    if (isRange_) {
      if (v > max_ || v < min_) {
        return kUnmappable;
      }
      return v - min_ + 1;
    }
    auto it = intIds_.find(v);
    if (it != intIds_.end()) {
      return it->second;
    }
    uint64_t id = numDistinct() + 1;
    intIds_.emplace(v, id);
    updateRange(v);
    if (numDistinct() >= rangeSize_) {
      return kUnmappable;
    }
    return id;
  }
  // valueId(StringView) (VectorHasher.h:707-738).
  uint64_t valueIdStr(const char* data, uint32_t size) {
    if (isRange_) {
      if (size > kStringASRangeMaxSize) {
        return kUnmappable;
      }
      int64_t number = stringAsNumber(data, size);
      if (number < min_ || number > max_) {
        return kUnmappable;
      }
      return number - min_ + 1;
    }
    std::string key(data, size);
    auto it = strIds_.find(key);
    if (it != strIds_.end()) {
      return it->second;
    }
    uint64_t id = numDistinct() + 1;
    strIds_.emplace(std::move(key), id);
    if (!rangeOverflow_) {
      if (size > kStringASRangeMaxSize) {
        setRangeOverflow();
      } else {
        updateRange(stringAsNumber(data, size));
      }
    }
    if (numDistinct() >= rangeSize_ || distinctOverflow_) {
      return kUnmappable;
    }
    return id;
  }
  // lookupValueId (VectorHasher.h:582-598, :740-768).
  uint64_t lookupInt64(int64_t v) const {
    if (isRange_) {
      if (v > max_ || v < min_) {
        return kUnmappable;
      }
      return v - min_ + 1;
    }
    auto it = intIds_.find(v);
    return it == intIds_.end() ? kUnmappable : it->second;
  }
  uint64_t lookupStr(const char* data, uint32_t size) const {
    if (isRange_) {
      if (size > kStringASRangeMaxSize) {
        return kUnmappable;
      }
      int64_t number = stringAsNumber(data, size);
      if (number < min_ || number > max_) {
        return kUnmappable;
      }
      return number - min_ + 1;
    }
    auto it = strIds_.find(std::string(data, size));
    return it == strIds_.end() ? kUnmappable : it->second;
  }
  // analyzeValue (VectorHasher.h:498-512, VectorHasher.cpp:661-686).
  void analyzeInt64(int64_t v) {
    if (!rangeOverflow_) {
      updateRange(v);
    }
    if (!distinctOverflow_) {
      if (intIds_.emplace(v, numDistinct() + 1).second) {
        if (numDistinct() > static_cast<size_t>(kMaxDistinct)) {
          setDistinctOverflow();
        }
      }
    }
  }
  void analyzeStr(const char* data, uint32_t size) {
    if (!rangeOverflow_) {
      if (size > kStringASRangeMaxSize) {
        setRangeOverflow();
      } else {
        updateRange(stringAsNumber(data, size));
      }
    }
    if (!distinctOverflow_) {
      if (strIds_.emplace(std::string(data, size), numDistinct() + 1).second) {
        if (numDistinct() > static_cast<size_t>(kMaxDistinct)) {
          setDistinctOverflow();
        }
      }
    }
  }

  bool isString() const { return kind_ == VX355_VARCHAR || kind_ == VX355_VARBINARY; }

  uint64_t valueIdAt(const Decoded& d, int32_t row) {
    if (kind_ == VX355_BOOLEAN) {
      return d.int64At(row) ? 2 : 1;  // VectorHasher.h:770-773
    }
    if (isString()) {
      uint8_t tmp;
      auto* sv = static_cast<const StringView*>(d.valuePtr(row, &tmp));
      return valueIdStr(sv->data(), sv->size);
    }
    if (kind_ == VX355_TIMESTAMP) {
      // VectorHasher.h:775-785
      uint8_t tmp;
      auto* ts = static_cast<const Timestamp*>(d.valuePtr(row, &tmp));
      if (ts->nanos % 1000000 != 0) {
        setRangeOverflow();
        setDistinctOverflow();
        return kUnmappable;
      }
      return valueIdInt64(ts->seconds * 1000 + static_cast<int64_t>(ts->nanos / 1000000));
    }
    return valueIdInt64(d.int64At(row));
  }
  uint64_t lookupValueIdAt(const Decoded& d, int32_t row) const {
    if (kind_ == VX355_BOOLEAN) {
      return d.int64At(row) ? 2 : 1;
    }
    if (isString()) {
      uint8_t tmp;
      auto* sv = static_cast<const StringView*>(d.valuePtr(row, &tmp));
      return lookupStr(sv->data(), sv->size);
    }
    if (kind_ == VX355_TIMESTAMP) {
      uint8_t tmp;
      auto* ts = static_cast<const Timestamp*>(d.valuePtr(row, &tmp));
      if (ts->nanos % 1000000 != 0) {
        return kUnmappable;
      }
      return lookupInt64(ts->seconds * 1000 + static_cast<int64_t>(ts->nanos / 1000000));
    }
    return lookupInt64(d.int64At(row));
  }
  void analyzeAt(const Decoded& d, int32_t row) {
    if (kind_ == VX355_BOOLEAN) {
      return;
    }
    if (isString()) {
      uint8_t tmp;
      auto* sv = static_cast<const StringView*>(d.valuePtr(row, &tmp));
      analyzeStr(sv->data(), sv->size);
      return;
    }
    if (kind_ == VX355_TIMESTAMP) {
      uint8_t tmp;
      auto* ts = static_cast<const Timestamp*>(d.valuePtr(row, &tmp));
      analyzeInt64(ts->seconds * 1000 + static_cast<int64_t>(ts->nanos / 1000000));
      return;
    }
    analyzeInt64(d.int64At(row));
  }

  int32_t kind_;
  uint64_t rangeSize_ = 0;
  uint64_t multiplier_ = 1;
  bool isRange_ = false;
  bool hasRange_ = false;
  bool rangeOverflow_ = false;
  bool distinctOverflow_ = false;
  int64_t min_ = 1;
  int64_t max_ = 0;
  // F14FastSet<UniqueValue> with ids in insertion order (VectorHasher.h:667-669).
  std::unordered_map<int64_t, uint64_t> intIds_;
  std::unordered_map<std::string, uint64_t> strIds_;
};

}  // namespace orc
