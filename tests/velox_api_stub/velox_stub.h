// TEST INFRASTRUCTURE - a stand-in for the slice of the Velox API that shim/*.cpp touches.
//
// This repository has no Velox to compile against (folly / fmt / glog / xsimd are absent from the image,
// /root/reference cannot be built here), so until round 5 no compiler had ever seen shim/*.cpp. This
// header declares - with the reference's names, signatures and member meanings - the ~40 classes and free
// functions the shim uses, with just enough behaviour behind them (flat / dictionary / constant / row
// vectors over plain buffers, a promise / future pair, a Driver that owns a list of operators) that the
// shim can be COMPILED (tests/test_shim_syntax.py: g++ -fsyntax-only) and RUN (tests/cpp/shim_plan_test.cpp on
// the CPU: plan translation, ROW flattening; tests/cpp/shim_operator_test.cpp on the GPU: the reference's
// Q1 plan shape through Vx355HashAggregation against the oracle).
//
// Signatures follow (reference paths relative to /root/reference/velox):
//   type/Type.h, type/Variant.h, type/StringView.h, buffer/Buffer.h, vector/BaseVector.h:88-756,
//   vector/FlatVector.h, vector/ComplexVector.h:30-145, vector/DecodedVector.h:79-310,
//   core/ITypedExpr.h:69-148, core/Expressions.h:61-600, core/PlanNode.h:671-925,1120-1370,3078-3560,
//   core/QueryConfig.h, exec/Operator.h:120-709, exec/Driver.h:233-273,789-847, exec/Task.h:588-593,
//   exec/OperatorUtils.h:71-75, common/future/VeloxPromise.h.
// Nothing here is product code and nothing in velox_amd/, include/ or shim/ includes it: the shim includes
// "velox/..." paths, which tests/velox_api_stub/velox/... forward to this file.
#pragma once

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <string_view>
#include <limits>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace facebook::velox {

using vector_size_t = int32_t;
using column_index_t = uint32_t;
constexpr column_index_t kConstantChannel = std::numeric_limits<column_index_t>::max();

// ---- common/base/Exceptions.h ------------------------------------------------------------------

class VeloxException : public std::runtime_error {
 public:
  explicit VeloxException(const std::string& message, std::string errorCode = "")
      : std::runtime_error(message), errorCode_(std::move(errorCode)) {}
  const std::string& errorCode() const { return errorCode_; }  // common/base/VeloxException.h:236

 private:
  std::string errorCode_;
};
namespace error_code {
inline constexpr const char* kMemCapExceeded = "MEM_CAP_EXCEEDED";  // common/base/VeloxException.h:123
}
class VeloxUserError : public VeloxException {
 public:
  using VeloxException::VeloxException;
};
class VeloxRuntimeError : public VeloxException {
 public:
  using VeloxException::VeloxException;
};

namespace detail {
inline void formatInto(std::ostringstream& out, const char* fmt) {
  out << fmt;
}
template <typename T, typename... Rest>
void formatInto(std::ostringstream& out, const char* fmt, const T& value, const Rest&... rest) {
  const char* brace = std::strstr(fmt, "{}");
  if (brace == nullptr) {
    out << fmt;
    return;
  }
  out.write(fmt, brace - fmt);
  out << value;
  formatInto(out, brace + 2, rest...);
}
inline std::string format() {
  return "";
}
template <typename... Args>
std::string format(const char* fmt, const Args&... args) {
  std::ostringstream out;
  formatInto(out, fmt, args...);
  return out.str();
}
}  // namespace detail

#define VELOX_FAIL(...) throw ::facebook::velox::VeloxRuntimeError(::facebook::velox::detail::format(__VA_ARGS__))
#define VELOX_USER_FAIL(...) throw ::facebook::velox::VeloxUserError(::facebook::velox::detail::format(__VA_ARGS__))
// common/memory/MemoryArbitrator.h:33-39: a VeloxRuntimeError with error code kMemCapExceeded
#define VELOX_MEM_POOL_CAP_EXCEEDED(...) \
  throw ::facebook::velox::VeloxRuntimeError(::facebook::velox::detail::format(__VA_ARGS__), ::facebook::velox::error_code::kMemCapExceeded)
#define VELOX_NYI(...) throw ::facebook::velox::VeloxRuntimeError("not yet implemented " + ::facebook::velox::detail::format(__VA_ARGS__))
#define VELOX_UNSUPPORTED(...) throw ::facebook::velox::VeloxUserError(::facebook::velox::detail::format(__VA_ARGS__))
#define VELOX_CHECK(cond, ...)                                                                         \
  do {                                                                                                 \
    if (!(cond)) {                                                                                     \
      throw ::facebook::velox::VeloxRuntimeError(std::string("check failed: " #cond " ") +            \
                                                 ::facebook::velox::detail::format(__VA_ARGS__));      \
    }                                                                                                  \
  } while (0)
#define VELOX_USER_CHECK(cond, ...)                                                                    \
  do {                                                                                                 \
    if (!(cond)) {                                                                                     \
      throw ::facebook::velox::VeloxUserError(std::string("check failed: " #cond " ") +               \
                                              ::facebook::velox::detail::format(__VA_ARGS__));         \
    }                                                                                                  \
  } while (0)
#define VELOX_CHECK_EQ(a, b, ...) VELOX_CHECK((a) == (b), __VA_ARGS__)
#define VELOX_CHECK_LT(a, b, ...) VELOX_CHECK((a) < (b), __VA_ARGS__)
#define VELOX_CHECK_LE(a, b, ...) VELOX_CHECK((a) <= (b), __VA_ARGS__)
#define VELOX_CHECK_GE(a, b, ...) VELOX_CHECK((a) >= (b), __VA_ARGS__)
#define VELOX_CHECK_NOT_NULL(p, ...) VELOX_CHECK((p) != nullptr, __VA_ARGS__)
#define VELOX_CHECK_NULL(p, ...) VELOX_CHECK((p) == nullptr, __VA_ARGS__)
#define VELOX_USER_CHECK_EQ(a, b, ...) VELOX_USER_CHECK((a) == (b), __VA_ARGS__)

// ---- common/future/VeloxPromise.h (folly::SemiFuture<folly::Unit> / folly::Promise) -------------

namespace detail {
struct FutureState {
  std::mutex m;
  std::condition_variable cv;
  bool done{false};
};
}  // namespace detail

class ContinueFuture {
 public:
  ContinueFuture() = default;
  explicit ContinueFuture(std::shared_ptr<detail::FutureState> state) : state_(std::move(state)) {}
  static ContinueFuture makeEmpty() {
    return ContinueFuture();
  }
  bool valid() const {
    return state_ != nullptr;
  }
  bool isReady() const {
    std::lock_guard<std::mutex> l(state_->m);
    return state_->done;
  }
  void wait() {
    std::unique_lock<std::mutex> l(state_->m);
    state_->cv.wait(l, [&] { return state_->done; });
  }

 private:
  std::shared_ptr<detail::FutureState> state_;
};

class ContinuePromise {
 public:
  ContinuePromise() : state_(std::make_shared<detail::FutureState>()) {}
  explicit ContinuePromise(std::string_view /*context*/) : ContinuePromise() {}
  ContinuePromise(ContinuePromise&&) = default;
  ContinuePromise& operator=(ContinuePromise&&) = default;
  ContinueFuture getSemiFuture() {
    return ContinueFuture(state_);
  }
  void setValue() {
    std::lock_guard<std::mutex> l(state_->m);
    state_->done = true;
    state_->cv.notify_all();
  }

 private:
  std::shared_ptr<detail::FutureState> state_;
};

// ---- type/Type.h -------------------------------------------------------------------------------

enum class TypeKind : int8_t {
  BOOLEAN = 0,
  TINYINT = 1,
  SMALLINT = 2,
  INTEGER = 3,
  BIGINT = 4,
  REAL = 5,
  DOUBLE = 6,
  VARCHAR = 7,
  VARBINARY = 8,
  TIMESTAMP = 9,
  HUGEINT = 10,
  ARRAY = 30,
  MAP = 31,
  ROW = 32,
  UNKNOWN = 33,
  FUNCTION = 34,
  OPAQUE = 35,
  INVALID = 36
};

class Type;
class RowType;
using TypePtr = std::shared_ptr<const Type>;
using RowTypePtr = std::shared_ptr<const RowType>;

class Type {
 public:
  explicit Type(TypeKind kind, bool date = false) : kind_(kind), date_(date) {}
  virtual ~Type() = default;
  TypeKind kind() const {
    return kind_;
  }
  bool isRow() const {
    return kind_ == TypeKind::ROW;
  }
  bool isBoolean() const {
    return kind_ == TypeKind::BOOLEAN;
  }
  bool isDate() const {
    return date_;
  }
  bool isPrimitiveType() const {
    return kind_ < TypeKind::ARRAY;
  }
  virtual uint32_t size() const {
    return 0;
  }
  virtual const TypePtr& childAt(uint32_t /*idx*/) const {
    VELOX_FAIL("scalar type has no children");
  }
  size_t cppSizeInBytes() const {
    switch (kind_) {
      case TypeKind::BOOLEAN:
      case TypeKind::TINYINT:
        return 1;
      case TypeKind::SMALLINT:
        return 2;
      case TypeKind::INTEGER:
      case TypeKind::REAL:
        return 4;
      case TypeKind::BIGINT:
      case TypeKind::DOUBLE:
        return 8;
      case TypeKind::VARCHAR:
      case TypeKind::VARBINARY:
      case TypeKind::TIMESTAMP:
      case TypeKind::HUGEINT:
        return 16;
      default:
        VELOX_FAIL("cppSizeInBytes of a complex type");
    }
  }
  const RowType& asRow() const;
  virtual std::string toString() const {
    static const char* names[] = {"BOOLEAN", "TINYINT", "SMALLINT", "INTEGER", "BIGINT", "REAL",
                                  "DOUBLE", "VARCHAR", "VARBINARY", "TIMESTAMP", "HUGEINT"};
    return date_ ? "DATE" : (kind_ <= TypeKind::HUGEINT ? names[static_cast<int>(kind_)] : "COMPLEX");
  }

 private:
  const TypeKind kind_;
  const bool date_;
};

class RowType : public Type {
 public:
  RowType(std::vector<std::string> names, std::vector<TypePtr> types)
      : Type(TypeKind::ROW), names_(std::move(names)), children_(std::move(types)) {}
  uint32_t size() const override {
    return static_cast<uint32_t>(children_.size());
  }
  const TypePtr& childAt(uint32_t idx) const override {
    VELOX_CHECK_LT(idx, children_.size());
    return children_[idx];
  }
  const std::vector<TypePtr>& children() const {
    return children_;
  }
  const std::vector<std::string>& names() const {
    return names_;
  }
  const std::string& nameOf(uint32_t idx) const {
    return names_.at(idx);
  }
  std::optional<uint32_t> getChildIdxIfExists(const std::string& name) const {
    for (uint32_t i = 0; i < names_.size(); ++i) {
      if (names_[i] == name) {
        return i;
      }
    }
    return std::nullopt;
  }
  uint32_t getChildIdx(const std::string& name) const {
    auto idx = getChildIdxIfExists(name);
    VELOX_USER_CHECK(idx.has_value(), "Field not found: {}", name);
    return *idx;
  }
  const TypePtr& findChild(const std::string& name) const {
    return children_[getChildIdx(name)];
  }
  std::string toString() const override {
    std::string out = "ROW<";
    for (size_t i = 0; i < names_.size(); ++i) {
      out += (i ? "," : "") + names_[i] + ":" + children_[i]->toString();
    }
    return out + ">";
  }

 private:
  const std::vector<std::string> names_;
  const std::vector<TypePtr> children_;
};

inline const RowType& Type::asRow() const {
  return dynamic_cast<const RowType&>(*this);
}

#define VX_STUB_SCALAR_TYPE(NAME, KIND)                                  \
  inline TypePtr NAME() {                                                \
    static const TypePtr t = std::make_shared<const Type>(TypeKind::KIND); \
    return t;                                                            \
  }
VX_STUB_SCALAR_TYPE(BOOLEAN, BOOLEAN)
VX_STUB_SCALAR_TYPE(TINYINT, TINYINT)
VX_STUB_SCALAR_TYPE(SMALLINT, SMALLINT)
VX_STUB_SCALAR_TYPE(INTEGER, INTEGER)
VX_STUB_SCALAR_TYPE(BIGINT, BIGINT)
VX_STUB_SCALAR_TYPE(REAL, REAL)
VX_STUB_SCALAR_TYPE(DOUBLE, DOUBLE)
VX_STUB_SCALAR_TYPE(VARCHAR, VARCHAR)
VX_STUB_SCALAR_TYPE(VARBINARY, VARBINARY)
VX_STUB_SCALAR_TYPE(TIMESTAMP, TIMESTAMP)
#undef VX_STUB_SCALAR_TYPE
inline TypePtr DATE() {
  static const TypePtr t = std::make_shared<const Type>(TypeKind::INTEGER, true);
  return t;
}
inline RowTypePtr ROW(std::vector<std::string> names, std::vector<TypePtr> types) {
  return std::make_shared<const RowType>(std::move(names), std::move(types));
}
inline RowTypePtr ROW(std::vector<TypePtr> types) {
  std::vector<std::string> names(types.size());
  return std::make_shared<const RowType>(std::move(names), std::move(types));
}
inline RowTypePtr asRowType(const TypePtr& type) {
  return std::dynamic_pointer_cast<const RowType>(type);
}

// ---- type/StringView.h:76-77 -------------------------------------------------------------------

struct StringView {
  static constexpr size_t kInlineSize = 12;
  StringView() {
    std::memset(this, 0, sizeof(*this));
  }
  StringView(const char* data, size_t len) {
    std::memset(this, 0, sizeof(*this));
    size_ = static_cast<uint32_t>(len);
    if (len <= kInlineSize) {
      std::memcpy(prefix_, data, len);
    } else {
      std::memcpy(prefix_, data, 4);
      value_.data = data;
    }
  }
  explicit StringView(const std::string& s) : StringView(s.data(), s.size()) {}
  bool isInline() const {
    return size_ <= kInlineSize;
  }
  uint32_t size() const {
    return size_;
  }
  const char* data() const {
    return isInline() ? prefix_ : value_.data;
  }
  std::string str() const {
    return std::string(data(), size_);
  }

 private:
  uint32_t size_;
  char prefix_[4];
  union {
    char inlined[8];
    const char* data;
  } value_;
};
static_assert(sizeof(StringView) == 16, "StringView is 16 bytes");

struct Timestamp {
  int64_t seconds{0};
  uint64_t nanos{0};
};

// ---- type/Variant.h ----------------------------------------------------------------------------

template <TypeKind K>
struct TypeTraits;
#define VX_STUB_TRAIT(KIND, T) \
  template <>                  \
  struct TypeTraits<TypeKind::KIND> { using NativeType = T; };
VX_STUB_TRAIT(BOOLEAN, bool)
VX_STUB_TRAIT(TINYINT, int8_t)
VX_STUB_TRAIT(SMALLINT, int16_t)
VX_STUB_TRAIT(INTEGER, int32_t)
VX_STUB_TRAIT(BIGINT, int64_t)
VX_STUB_TRAIT(REAL, float)
VX_STUB_TRAIT(DOUBLE, double)
VX_STUB_TRAIT(VARCHAR, std::string)
VX_STUB_TRAIT(VARBINARY, std::string)
#undef VX_STUB_TRAIT

class Variant {
 public:
  Variant() : kind_(TypeKind::UNKNOWN), null_(true) {}
  Variant(bool v) : kind_(TypeKind::BOOLEAN), null_(false), i_(v) {}
  Variant(int8_t v) : kind_(TypeKind::TINYINT), null_(false), i_(v) {}
  Variant(int16_t v) : kind_(TypeKind::SMALLINT), null_(false), i_(v) {}
  Variant(int32_t v) : kind_(TypeKind::INTEGER), null_(false), i_(v) {}
  Variant(int64_t v) : kind_(TypeKind::BIGINT), null_(false), i_(v) {}
  Variant(float v) : kind_(TypeKind::REAL), null_(false), d_(v) {}
  Variant(double v) : kind_(TypeKind::DOUBLE), null_(false), d_(v) {}
  Variant(const char* v) : kind_(TypeKind::VARCHAR), null_(false), s_(v) {}
  Variant(std::string v) : kind_(TypeKind::VARCHAR), null_(false), s_(std::move(v)) {}
  static Variant null(TypeKind kind) {
    Variant v;
    v.kind_ = kind;
    return v;
  }
  TypeKind kind() const {
    return kind_;
  }
  bool isNull() const {
    return null_;
  }
  bool hasValue() const {
    return !null_;
  }
  template <TypeKind K>
  typename TypeTraits<K>::NativeType value() const {
    VELOX_CHECK(K == kind_ && !null_, "variant kind mismatch");
    using T = typename TypeTraits<K>::NativeType;
    if constexpr (std::is_same_v<T, std::string>) {
      return s_;
    } else if constexpr (std::is_floating_point_v<T>) {
      return static_cast<T>(d_);
    } else {
      return static_cast<T>(i_);
    }
  }

 private:
  TypeKind kind_;
  bool null_;
  int64_t i_{0};
  double d_{0};
  std::string s_;
};
using variant = Variant;

// ---- common/memory/Memory.h, buffer/Buffer.h ---------------------------------------------------

namespace memory {
class MemoryPool {
 public:
  int64_t reservedBytes() const {
    return 0;
  }
};
}  // namespace memory

namespace bits {
inline bool isBitSet(const uint64_t* bits, int32_t idx) {
  return (bits[idx >> 6] >> (idx & 63)) & 1;
}
inline void setBit(uint64_t* bits, int32_t idx, bool value = true) {
  if (value) {
    bits[idx >> 6] |= 1ULL << (idx & 63);
  } else {
    bits[idx >> 6] &= ~(1ULL << (idx & 63));
  }
}
inline void clearBit(uint64_t* bits, int32_t idx) {
  setBit(bits, idx, false);
}
inline bool isBitNull(const uint64_t* bits, int32_t idx) {
  return !isBitSet(bits, idx);  // common/base/Nulls.h:26-38: 1 = not null
}
inline uint64_t nwords(int32_t bits) {
  return (static_cast<uint64_t>(bits) + 63) / 64;
}
constexpr uint64_t kNotNull64 = ~0ULL;
}  // namespace bits

class Buffer {
 public:
  explicit Buffer(size_t bytes, uint8_t fill = 0) : data_(bytes + 64, fill), size_(bytes) {}
  template <typename T>
  const T* as() const {
    return reinterpret_cast<const T*>(data_.data());
  }
  template <typename T>
  T* asMutable() {
    return reinterpret_cast<T*>(data_.data());
  }
  size_t size() const {
    return size_;
  }
  size_t capacity() const {
    return data_.size();
  }
  void setSize(size_t bytes) {
    if (bytes + 64 > data_.size()) {
      data_.resize(bytes + 64);
    }
    size_ = bytes;
  }

 private:
  std::vector<uint8_t> data_;
  size_t size_;
};
using BufferPtr = std::shared_ptr<Buffer>;

struct AlignedBuffer {
  template <typename T>
  static BufferPtr allocate(size_t numElements, memory::MemoryPool* /*pool*/, const std::optional<T>& init = std::nullopt) {
    size_t bytes = std::is_same_v<T, bool> ? (numElements + 7) / 8 : numElements * sizeof(T);
    auto b = std::make_shared<Buffer>(bytes);
    if (init.has_value()) {
      if constexpr (std::is_same_v<T, bool>) {
        std::memset(b->template asMutable<uint8_t>(), *init ? 0xff : 0, bytes);
      } else {
        std::fill_n(b->template asMutable<T>(), numElements, *init);
      }
    }
    return b;
  }
};

inline BufferPtr allocateIndices(vector_size_t size, memory::MemoryPool* pool) {
  return AlignedBuffer::allocate<vector_size_t>(size, pool, 0);
}
inline BufferPtr allocateNulls(vector_size_t size, memory::MemoryPool* pool, bool initValue = true) {
  return AlignedBuffer::allocate<bool>(bits::nwords(size) * 64, pool, initValue);
}

// ---- vector/BaseVector.h, FlatVector.h, ComplexVector.h, DictionaryVector.h, ConstantVector.h --

namespace VectorEncoding {
enum class Simple { BIASED, CONSTANT, DICTIONARY, FLAT, SEQUENCE, ROW, MAP, ARRAY, LAZY, FUNCTION };
}

class BaseVector;
using VectorPtr = std::shared_ptr<BaseVector>;
template <typename T>
class FlatVector;
class RowVector;
using RowVectorPtr = std::shared_ptr<RowVector>;

class BaseVector {
 public:
  BaseVector(memory::MemoryPool* pool, TypePtr type, VectorEncoding::Simple encoding, BufferPtr nulls, vector_size_t length)
      : pool_(pool), type_(std::move(type)), encoding_(encoding), nulls_(std::move(nulls)), length_(length) {}
  virtual ~BaseVector() = default;

  VectorEncoding::Simple encoding() const {
    return encoding_;
  }
  bool isConstantEncoding() const {
    return encoding_ == VectorEncoding::Simple::CONSTANT;
  }
  const TypePtr& type() const {
    return type_;
  }
  TypeKind typeKind() const {
    return type_->kind();
  }
  vector_size_t size() const {
    return length_;
  }
  memory::MemoryPool* pool() const {
    return pool_;
  }
  virtual bool mayHaveNulls() const {
    return nulls_ != nullptr;
  }
  virtual bool isNullAt(vector_size_t idx) const {
    return nulls_ != nullptr && bits::isBitNull(nulls_->as<uint64_t>(), idx);
  }
  const BufferPtr& nulls() const {
    return nulls_;
  }
  const uint64_t* rawNulls() const {
    return nulls_ ? nulls_->as<uint64_t>() : nullptr;
  }
  uint64_t* mutableRawNulls() {
    ensureNulls();
    return nulls_->asMutable<uint64_t>();
  }
  virtual void setNull(vector_size_t idx, bool isNull) {
    if (nulls_ == nullptr && !isNull) {
      return;
    }
    ensureNulls();
    bits::setBit(nulls_->asMutable<uint64_t>(), idx, !isNull);
  }
  virtual void resize(vector_size_t newSize, bool /*setNotNull*/ = true) {
    if (nulls_ != nullptr && bits::nwords(newSize) * 8 > nulls_->size()) {
      const auto old = nulls_->size();
      nulls_->setSize(bits::nwords(newSize) * 8);
      std::memset(nulls_->asMutable<uint8_t>() + old, 0xff, nulls_->size() - old);
    }
    length_ = newSize;
  }
  virtual BaseVector* loadedVector() {
    return this;
  }
  virtual const BaseVector* loadedVector() const {
    return this;
  }
  // vector/BaseVector.h:716-718: only flat vectors of scalars have a values buffer
  virtual const BufferPtr& values() const {
    VELOX_UNSUPPORTED("Only flat vectors have a values buffer");
  }
  virtual const BaseVector* wrappedVector() const {
    return this;
  }
  virtual vector_size_t wrappedIndex(vector_size_t idx) const {
    return idx;
  }

  template <typename T>
  T* as() {
    return dynamic_cast<T*>(this);
  }
  template <typename T>
  const T* as() const {
    return dynamic_cast<const T*>(this);
  }
  template <typename T>
  FlatVector<T>* asFlatVector() {
    return dynamic_cast<FlatVector<T>*>(this);
  }
  template <typename T>
  const FlatVector<T>* asFlatVector() const {
    return dynamic_cast<const FlatVector<T>*>(this);
  }

  static VectorPtr create(const TypePtr& type, vector_size_t size, memory::MemoryPool* pool);
  template <typename T>
  static std::shared_ptr<T> create(const TypePtr& type, vector_size_t size, memory::MemoryPool* pool) {
    return std::static_pointer_cast<T>(create(type, size, pool));
  }
  static VectorPtr createNullConstant(const TypePtr& type, vector_size_t size, memory::MemoryPool* pool);
  static VectorPtr wrapInDictionary(BufferPtr nulls, BufferPtr indices, vector_size_t size, VectorPtr vector);
  static VectorPtr wrapInConstant(vector_size_t length, vector_size_t index, VectorPtr vector);
  static void flattenVector(VectorPtr& vector);

 protected:
  void ensureNulls() {
    if (nulls_ == nullptr) {
      nulls_ = allocateNulls(std::max(length_, 1), pool_, true);
    }
  }
  memory::MemoryPool* pool_;
  TypePtr type_;
  VectorEncoding::Simple encoding_;
  BufferPtr nulls_;
  vector_size_t length_;
};

template <typename T>
class FlatVector : public BaseVector {
 public:
  FlatVector(memory::MemoryPool* pool, const TypePtr& type, BufferPtr nulls, vector_size_t length, BufferPtr values,
             std::vector<BufferPtr> stringBuffers = {})
      : BaseVector(pool, type, VectorEncoding::Simple::FLAT, std::move(nulls), length),
        values_(std::move(values)),
        stringBuffers_(std::move(stringBuffers)) {}
  const BufferPtr& values() const override {
    return values_;
  }
  const T* rawValues() const {
    return values_->as<T>();
  }
  T* mutableRawValues() {
    return values_->asMutable<T>();
  }
  T valueAt(vector_size_t idx) const {
    if constexpr (std::is_same_v<T, bool>) {
      return bits::isBitSet(values_->as<uint64_t>(), idx);
    } else {
      return values_->as<T>()[idx];
    }
  }
  void set(vector_size_t idx, T value) {
    if constexpr (std::is_same_v<T, bool>) {
      bits::setBit(values_->asMutable<uint64_t>(), idx, value);
    } else if constexpr (std::is_same_v<T, StringView>) {
      // FlatVector<StringView>::set copies non-inline bytes into a string buffer of the vector
      if (!value.isInline()) {
        auto buf = std::make_shared<Buffer>(value.size());
        std::memcpy(buf->template asMutable<char>(), value.data(), value.size());
        stringBuffers_.push_back(buf);
        value = StringView(buf->template as<char>(), value.size());
      }
      values_->asMutable<T>()[idx] = value;
    } else {
      values_->asMutable<T>()[idx] = value;
    }
    if (nulls_ != nullptr) {
      setNull(idx, false);
    }
  }
  void resize(vector_size_t newSize, bool setNotNull = true) override {
    const size_t bytes = std::is_same_v<T, bool> ? bits::nwords(newSize) * 8 : static_cast<size_t>(newSize) * sizeof(T);
    if (bytes > values_->size()) {
      values_->setSize(bytes);
    }
    BaseVector::resize(newSize, setNotNull);
  }
  const std::vector<BufferPtr>& stringBuffers() const {
    return stringBuffers_;
  }

 private:
  BufferPtr values_;
  std::vector<BufferPtr> stringBuffers_;
};

class RowVector : public BaseVector {
 public:
  RowVector(memory::MemoryPool* pool, const TypePtr& type, BufferPtr nulls, vector_size_t length,
            std::vector<VectorPtr> children, std::optional<vector_size_t> /*nullCount*/ = std::nullopt)
      : BaseVector(pool, type, VectorEncoding::Simple::ROW, std::move(nulls), length), children_(std::move(children)) {}
  size_t childrenSize() const {
    return children_.size();
  }
  VectorPtr& childAt(column_index_t idx) {
    VELOX_CHECK_LT(idx, children_.size());
    return children_[idx];
  }
  const VectorPtr& childAt(column_index_t idx) const {
    VELOX_CHECK_LT(idx, children_.size());
    return children_[idx];
  }
  std::vector<VectorPtr>& children() {
    return children_;
  }
  const std::vector<VectorPtr>& children() const {
    return children_;
  }
  void resize(vector_size_t newSize, bool setNotNull = true) override {
    for (auto& child : children_) {
      if (child != nullptr) {
        child->resize(newSize, setNotNull);
      }
    }
    BaseVector::resize(newSize, setNotNull);
  }

 private:
  std::vector<VectorPtr> children_;
};

class DictionaryVectorBase : public BaseVector {
 public:
  DictionaryVectorBase(memory::MemoryPool* pool, BufferPtr nulls, vector_size_t length, VectorPtr base, BufferPtr indices)
      : BaseVector(pool, base->type(), VectorEncoding::Simple::DICTIONARY, std::move(nulls), length),
        base_(std::move(base)),
        indices_(std::move(indices)) {}
  const VectorPtr& valueVector() const {
    return base_;
  }
  const BufferPtr& indices() const {
    return indices_;
  }
  bool isNullAt(vector_size_t idx) const override {
    return BaseVector::isNullAt(idx) || base_->isNullAt(indices_->as<vector_size_t>()[idx]);
  }
  bool mayHaveNulls() const override {
    return nulls_ != nullptr || base_->mayHaveNulls();
  }
  const BaseVector* wrappedVector() const override {
    return base_->wrappedVector();
  }
  vector_size_t wrappedIndex(vector_size_t idx) const override {
    return base_->wrappedIndex(indices_->as<vector_size_t>()[idx]);
  }

 private:
  VectorPtr base_;
  BufferPtr indices_;
};

class ConstantVectorBase : public BaseVector {
 public:
  // a constant over row 'index' of 'base' (wrapInConstant), or a null constant (base == nullptr)
  ConstantVectorBase(memory::MemoryPool* pool, TypePtr type, vector_size_t length, VectorPtr base, vector_size_t index)
      : BaseVector(pool, std::move(type), VectorEncoding::Simple::CONSTANT, nullptr, length), base_(std::move(base)), index_(index) {}
  bool isNullAt(vector_size_t /*idx*/) const override {
    return base_ == nullptr || base_->isNullAt(index_);
  }
  bool mayHaveNulls() const override {
    return isNullAt(0);
  }
  const VectorPtr& valueVector() const {
    return base_;
  }
  vector_size_t index() const {
    return index_;
  }
  const BaseVector* wrappedVector() const override {
    return base_ ? base_->wrappedVector() : this;
  }
  vector_size_t wrappedIndex(vector_size_t /*idx*/) const override {
    return base_ ? base_->wrappedIndex(index_) : 0;
  }

 private:
  VectorPtr base_;
  vector_size_t index_;
};

namespace detail {
template <typename T>
VectorPtr makeFlat(const TypePtr& type, vector_size_t size, memory::MemoryPool* pool) {
  const size_t bytes = std::is_same_v<T, bool> ? bits::nwords(size) * 8 : static_cast<size_t>(size) * sizeof(T);
  return std::make_shared<FlatVector<T>>(pool, type, nullptr, size, std::make_shared<Buffer>(bytes));
}
}  // namespace detail

inline VectorPtr BaseVector::create(const TypePtr& type, vector_size_t size, memory::MemoryPool* pool) {
  switch (type->kind()) {
    case TypeKind::BOOLEAN:
      return detail::makeFlat<bool>(type, size, pool);
    case TypeKind::TINYINT:
      return detail::makeFlat<int8_t>(type, size, pool);
    case TypeKind::SMALLINT:
      return detail::makeFlat<int16_t>(type, size, pool);
    case TypeKind::INTEGER:
      return detail::makeFlat<int32_t>(type, size, pool);
    case TypeKind::BIGINT:
      return detail::makeFlat<int64_t>(type, size, pool);
    case TypeKind::REAL:
      return detail::makeFlat<float>(type, size, pool);
    case TypeKind::DOUBLE:
      return detail::makeFlat<double>(type, size, pool);
    case TypeKind::VARCHAR:
    case TypeKind::VARBINARY:
      return detail::makeFlat<StringView>(type, size, pool);
    case TypeKind::TIMESTAMP:
      return detail::makeFlat<Timestamp>(type, size, pool);
    case TypeKind::ROW: {
      std::vector<VectorPtr> children;
      for (const auto& child : type->asRow().children()) {
        children.push_back(create(child, size, pool));
      }
      return std::make_shared<RowVector>(pool, type, nullptr, size, std::move(children));
    }
    default:
      VELOX_NYI("BaseVector::create of {}", type->toString());
  }
}

inline VectorPtr BaseVector::createNullConstant(const TypePtr& type, vector_size_t size, memory::MemoryPool* pool) {
  return std::make_shared<ConstantVectorBase>(pool, type, size, nullptr, 0);
}
inline VectorPtr BaseVector::wrapInDictionary(BufferPtr nulls, BufferPtr indices, vector_size_t size, VectorPtr vector) {
  auto* pool = vector->pool();
  return std::make_shared<DictionaryVectorBase>(pool, std::move(nulls), size, std::move(vector), std::move(indices));
}
inline VectorPtr BaseVector::wrapInConstant(vector_size_t length, vector_size_t index, VectorPtr vector) {
  auto* pool = vector->pool();
  auto type = vector->type();
  return std::make_shared<ConstantVectorBase>(pool, type, length, std::move(vector), index);
}

/// DecodedVector (vector/DecodedVector.h:79-310): any encoding reduced to base data + indices + nulls.
class DecodedVector {
 public:
  DecodedVector() = default;
  DecodedVector(const DecodedVector&) = delete;
  DecodedVector(DecodedVector&&) = default;
  explicit DecodedVector(const BaseVector& vector, bool /*loadLazy*/ = true) {
    decode(vector);
  }
  void decode(const BaseVector& vector, bool /*loadLazy*/ = true) {
    size_ = vector.size();
    base_ = vector.wrappedVector();
    isIdentityMapping_ = isConstantMapping_ = false;
    indicesHolder_.clear();
    nullsHolder_.clear();
    nulls_ = nullptr;
    switch (vector.encoding()) {
      case VectorEncoding::Simple::FLAT:
      case VectorEncoding::Simple::ROW:
        isIdentityMapping_ = true;
        nulls_ = vector.rawNulls();
        break;
      case VectorEncoding::Simple::CONSTANT:
        isConstantMapping_ = true;
        constantIndex_ = vector.wrappedIndex(0);
        constantNull_ = vector.isNullAt(0);
        break;
      default: {
        indicesHolder_.resize(size_);
        bool anyNull = false;
        for (vector_size_t i = 0; i < size_; ++i) {
          indicesHolder_[i] = vector.wrappedIndex(i);
          anyNull |= vector.isNullAt(i);
        }
        if (anyNull) {
          nullsHolder_.assign(bits::nwords(size_), bits::kNotNull64);
          for (vector_size_t i = 0; i < size_; ++i) {
            if (vector.isNullAt(i)) {
              bits::clearBit(nullsHolder_.data(), i);
            }
          }
          nulls_ = nullsHolder_.data();
        }
      }
    }
    data_ = (base_->encoding() == VectorEncoding::Simple::FLAT) ? base_->values()->as<void>() : nullptr;
  }
  template <typename T>
  const T* data() const {
    return reinterpret_cast<const T*>(data_);
  }
  const uint64_t* nulls(const void* /*rows*/ = nullptr) {
    if (isConstantMapping_ && constantNull_ && nullsHolder_.empty()) {
      nullsHolder_.assign(bits::nwords(std::max(size_, 1)), 0);
      nulls_ = nullsHolder_.data();
    }
    return nulls_;
  }
  const vector_size_t* indices() const {
    return indicesHolder_.data();
  }
  vector_size_t index(vector_size_t idx) const {
    return isIdentityMapping_ ? idx : (isConstantMapping_ ? constantIndex_ : indicesHolder_[idx]);
  }
  bool isNullAt(vector_size_t idx) const {
    if (isConstantMapping_) {
      return constantNull_;
    }
    return nulls_ != nullptr && bits::isBitNull(nulls_, idx);
  }
  bool mayHaveNulls() const {
    return nulls_ != nullptr || (isConstantMapping_ && constantNull_);
  }
  vector_size_t size() const {
    return size_;
  }
  const BaseVector* base() const {
    return base_;
  }
  bool isIdentityMapping() const {
    return isIdentityMapping_;
  }
  bool isConstantMapping() const {
    return isConstantMapping_;
  }

 private:
  const BaseVector* base_{nullptr};
  const void* data_{nullptr};
  const uint64_t* nulls_{nullptr};
  std::vector<vector_size_t> indicesHolder_;
  std::vector<uint64_t> nullsHolder_;
  vector_size_t size_{0};
  vector_size_t constantIndex_{0};
  bool constantNull_{false};
  bool isIdentityMapping_{false};
  bool isConstantMapping_{false};
};

inline void BaseVector::flattenVector(VectorPtr& vector) {
  if (vector == nullptr || vector->encoding() == VectorEncoding::Simple::FLAT) {
    return;
  }
  if (vector->encoding() == VectorEncoding::Simple::ROW) {
    for (auto& child : vector->as<RowVector>()->children()) {
      flattenVector(child);
    }
    return;
  }
  DecodedVector decoded(*vector);
  auto flat = create(vector->type(), vector->size(), vector->pool());
  if (vector->type()->isRow()) {
    // a wrapped struct: flatten every field through the wrapping
    const auto* base = decoded.base()->as<RowVector>();
    auto* row = flat->as<RowVector>();
    auto indices = allocateIndices(vector->size(), vector->pool());
    for (vector_size_t i = 0; i < vector->size(); ++i) {
      indices->asMutable<vector_size_t>()[i] = decoded.isNullAt(i) ? 0 : decoded.index(i);
      if (decoded.isNullAt(i) || base == nullptr || base->isNullAt(decoded.index(i))) {
        row->setNull(i, true);
      }
    }
    for (size_t c = 0; base != nullptr && c < base->childrenSize(); ++c) {
      VectorPtr wrapped = wrapInDictionary(nullptr, indices, vector->size(), base->childAt(c));
      flattenVector(wrapped);
      row->childAt(c) = wrapped;
    }
    vector = flat;
    return;
  }
  const size_t width = vector->typeKind() == TypeKind::BOOLEAN ? 0 : vector->type()->cppSizeInBytes();
  for (vector_size_t i = 0; i < vector->size(); ++i) {
    if (decoded.isNullAt(i)) {
      flat->setNull(i, true);
    } else if (width == 0) {
      flat->as<FlatVector<bool>>()->set(i, bits::isBitSet(decoded.data<uint64_t>(), decoded.index(i)));
    } else {
      std::memcpy(const_cast<uint8_t*>(flat->values()->as<uint8_t>()) + i * width,
                  decoded.data<uint8_t>() + static_cast<size_t>(decoded.index(i)) * width, width);
    }
  }
  vector = flat;
}

// ---- core/ITypedExpr.h, core/Expressions.h -----------------------------------------------------

namespace core {

using PlanNodeId = std::string;

enum class ExprKind : int32_t { kInput = 0, kFieldAccess = 1, kDereference = 2, kCall = 3, kCast = 4, kConstant = 5, kConcat = 6, kLambda = 7 };

class ITypedExpr;
using TypedExprPtr = std::shared_ptr<const ITypedExpr>;

class ITypedExpr {
 public:
  ITypedExpr(ExprKind kind, TypePtr type) : kind_(kind), type_(std::move(type)) {}
  ITypedExpr(ExprKind kind, TypePtr type, std::vector<TypedExprPtr> inputs)
      : kind_(kind), type_(std::move(type)), inputs_(std::move(inputs)) {}
  virtual ~ITypedExpr() = default;
  ExprKind kind() const {
    return kind_;
  }
  const TypePtr& type() const {
    return type_;
  }
  const std::vector<TypedExprPtr>& inputs() const {
    return inputs_;
  }
  bool isInputKind() const {
    return kind_ == ExprKind::kInput;
  }
  bool isFieldAccessKind() const {
    return kind_ == ExprKind::kFieldAccess;
  }
  bool isCallKind() const {
    return kind_ == ExprKind::kCall;
  }
  bool isCastKind() const {
    return kind_ == ExprKind::kCast;
  }
  bool isConstantKind() const {
    return kind_ == ExprKind::kConstant;
  }
  template <typename T>
  const T* asUnchecked() const {
    return dynamic_cast<const T*>(this);
  }
  virtual std::string toString() const = 0;

 private:
  ExprKind kind_;
  TypePtr type_;
  std::vector<TypedExprPtr> inputs_;
};

class InputTypedExpr : public ITypedExpr {
 public:
  explicit InputTypedExpr(TypePtr type) : ITypedExpr(ExprKind::kInput, std::move(type)) {}
  std::string toString() const override {
    return "ROW";
  }
};

class ConstantTypedExpr : public ITypedExpr {
 public:
  ConstantTypedExpr(TypePtr type, Variant value) : ITypedExpr(ExprKind::kConstant, std::move(type)), value_(std::move(value)) {}
  explicit ConstantTypedExpr(const VectorPtr& value)
      : ITypedExpr(ExprKind::kConstant, value->type()),
        valueVector_(value->isConstantEncoding() ? value : BaseVector::wrapInConstant(1, 0, value)) {}
  bool hasValueVector() const {
    return valueVector_ != nullptr;
  }
  const Variant& value() const {
    return value_;
  }
  const VectorPtr& valueVector() const {
    return valueVector_;
  }
  bool isNull() const {
    return hasValueVector() ? valueVector_->isNullAt(0) : value_.isNull();
  }
  std::string toString() const override {
    return "const";
  }

 private:
  Variant value_;
  VectorPtr valueVector_;
};

class CallTypedExpr : public ITypedExpr {
 public:
  CallTypedExpr(TypePtr type, std::vector<TypedExprPtr> inputs, std::string name)
      : ITypedExpr(ExprKind::kCall, std::move(type), std::move(inputs)), name_(std::move(name)) {}
  virtual const std::string& name() const {
    return name_;
  }
  std::string toString() const override {
    std::string out = name_ + "(";
    for (size_t i = 0; i < inputs().size(); ++i) {
      out += (i ? "," : "") + inputs()[i]->toString();
    }
    return out + ")";
  }

 private:
  std::string name_;
};
using CallTypedExprPtr = std::shared_ptr<const CallTypedExpr>;

class FieldAccessTypedExpr : public ITypedExpr {
 public:
  FieldAccessTypedExpr(TypePtr type, std::string name)
      : ITypedExpr(ExprKind::kFieldAccess, std::move(type)), name_(std::move(name)), isInputColumn_(true) {}
  FieldAccessTypedExpr(TypePtr type, TypedExprPtr input, std::string name)
      : ITypedExpr(ExprKind::kFieldAccess, std::move(type), {std::move(input)}),
        name_(std::move(name)),
        isInputColumn_(inputs()[0]->isInputKind()) {}
  const std::string& name() const {
    return name_;
  }
  bool isInputColumn() const {
    return isInputColumn_;
  }
  std::string toString() const override {
    return "\"" + name_ + "\"";
  }

 private:
  std::string name_;
  bool isInputColumn_;
};
using FieldAccessTypedExprPtr = std::shared_ptr<const FieldAccessTypedExpr>;

class CastTypedExpr : public ITypedExpr {
 public:
  CastTypedExpr(const TypePtr& type, const TypedExprPtr& input, bool isTryCast)
      : ITypedExpr(ExprKind::kCast, type, {input}), isTryCast_(isTryCast) {}
  bool isTryCast() const {
    return isTryCast_;
  }
  std::string toString() const override {
    return "cast(" + inputs()[0]->toString() + ")";
  }

 private:
  bool isTryCast_;
};

// ---- core/PlanNode.h ---------------------------------------------------------------------------

class PlanNode;
using PlanNodePtr = std::shared_ptr<const PlanNode>;

class PlanNode {
 public:
  explicit PlanNode(PlanNodeId id) : id_(std::move(id)) {}
  virtual ~PlanNode() = default;
  const PlanNodeId& id() const {
    return id_;
  }
  virtual const RowTypePtr& outputType() const = 0;
  virtual const std::vector<PlanNodePtr>& sources() const = 0;
  virtual std::string_view name() const = 0;

 private:
  const PlanNodeId id_;
};

/// A leaf with a given output type (core::ValuesNode / TableScanNode stand-in).
class ValuesNode : public PlanNode {
 public:
  ValuesNode(const PlanNodeId& id, RowTypePtr type) : PlanNode(id), type_(std::move(type)) {}
  const RowTypePtr& outputType() const override {
    return type_;
  }
  const std::vector<PlanNodePtr>& sources() const override {
    static const std::vector<PlanNodePtr> kEmpty;
    return kEmpty;
  }
  std::string_view name() const override {
    return "Values";
  }

 private:
  RowTypePtr type_;
};

class FilterNode : public PlanNode {
 public:
  FilterNode(const PlanNodeId& id, TypedExprPtr filter, PlanNodePtr source)
      : PlanNode(id), sources_{std::move(source)}, filter_(std::move(filter)) {}
  const RowTypePtr& outputType() const override {
    return sources_[0]->outputType();
  }
  const std::vector<PlanNodePtr>& sources() const override {
    return sources_;
  }
  const TypedExprPtr& filter() const {
    return filter_;
  }
  std::string_view name() const override {
    return "Filter";
  }

 private:
  const std::vector<PlanNodePtr> sources_;
  const TypedExprPtr filter_;
};

class ProjectNode : public PlanNode {
 public:
  ProjectNode(const PlanNodeId& id, const std::vector<std::string>& names, const std::vector<TypedExprPtr>& projections,
              PlanNodePtr source)
      : PlanNode(id), sources_{std::move(source)}, names_(names), projections_(projections) {
    std::vector<TypePtr> types;
    for (const auto& p : projections_) {
      types.push_back(p->type());
    }
    outputType_ = ROW(names_, std::move(types));
  }
  const RowTypePtr& outputType() const override {
    return outputType_;
  }
  const std::vector<PlanNodePtr>& sources() const override {
    return sources_;
  }
  const std::vector<std::string>& names() const {
    return names_;
  }
  const std::vector<TypedExprPtr>& projections() const {
    return projections_;
  }
  std::string_view name() const override {
    return "Project";
  }

 private:
  const std::vector<PlanNodePtr> sources_;
  const std::vector<std::string> names_;
  const std::vector<TypedExprPtr> projections_;
  RowTypePtr outputType_;
};

struct SortOrder {
  bool ascending{true};
  bool nullsFirst{false};
};

class AggregationNode : public PlanNode {
 public:
  enum class Step { kPartial, kFinal, kIntermediate, kSingle };
  struct Aggregate {
    CallTypedExprPtr call;
    std::vector<TypePtr> rawInputTypes;
    FieldAccessTypedExprPtr mask{};
    std::vector<FieldAccessTypedExprPtr> sortingKeys{};
    std::vector<SortOrder> sortingOrders{};
    bool distinct{false};
  };
  AggregationNode(const PlanNodeId& id, Step step, const std::vector<FieldAccessTypedExprPtr>& groupingKeys,
                  const std::vector<FieldAccessTypedExprPtr>& preGroupedKeys, const std::vector<std::string>& aggregateNames,
                  const std::vector<Aggregate>& aggregates, bool ignoreNullKeys, bool noGroupsSpanBatches, PlanNodePtr source)
      : PlanNode(id),
        step_(step),
        groupingKeys_(groupingKeys),
        preGroupedKeys_(preGroupedKeys),
        aggregateNames_(aggregateNames),
        aggregates_(aggregates),
        ignoreNullKeys_(ignoreNullKeys),
        noGroupsSpanBatches_(noGroupsSpanBatches),
        sources_{std::move(source)} {
    std::vector<std::string> names;
    std::vector<TypePtr> types;
    for (const auto& key : groupingKeys_) {
      names.push_back(key->name());
      types.push_back(key->type());
    }
    for (size_t i = 0; i < aggregates_.size(); ++i) {
      names.push_back(aggregateNames_[i]);
      types.push_back(aggregates_[i].call->type());
    }
    outputType_ = ROW(std::move(names), std::move(types));
  }
  Step step() const {
    return step_;
  }
  const std::vector<FieldAccessTypedExprPtr>& groupingKeys() const {
    return groupingKeys_;
  }
  const std::vector<FieldAccessTypedExprPtr>& preGroupedKeys() const {
    return preGroupedKeys_;
  }
  const std::vector<std::string>& aggregateNames() const {
    return aggregateNames_;
  }
  const std::vector<Aggregate>& aggregates() const {
    return aggregates_;
  }
  const std::vector<vector_size_t>& globalGroupingSets() const {
    return globalGroupingSets_;
  }
  std::optional<FieldAccessTypedExprPtr> groupId() const {
    return std::nullopt;
  }
  bool ignoreNullKeys() const {
    return ignoreNullKeys_;
  }
  bool noGroupsSpanBatches() const {
    return noGroupsSpanBatches_;
  }
  const RowTypePtr& outputType() const override {
    return outputType_;
  }
  const std::vector<PlanNodePtr>& sources() const override {
    return sources_;
  }
  std::string_view name() const override {
    return "Aggregation";
  }

 private:
  const Step step_;
  const std::vector<FieldAccessTypedExprPtr> groupingKeys_;
  const std::vector<FieldAccessTypedExprPtr> preGroupedKeys_;
  const std::vector<std::string> aggregateNames_;
  const std::vector<Aggregate> aggregates_;
  const std::vector<vector_size_t> globalGroupingSets_;
  const bool ignoreNullKeys_;
  const bool noGroupsSpanBatches_;
  const std::vector<PlanNodePtr> sources_;
  RowTypePtr outputType_;
};

enum class JoinType {
  kInner = 0,
  kLeft = 1,
  kRight = 2,
  kFull = 3,
  kLeftSemiFilter = 4,
  kCountingLeftSemiFilter = 5,
  kLeftSemiProject = 6,
  kRightSemiFilter = 7,
  kRightSemiProject = 8,
  kAnti = 9,
  kCountingAnti = 10,
  kRightAnti = 11,
};

class HashJoinNode : public PlanNode {
 public:
  HashJoinNode(const PlanNodeId& id, JoinType joinType, bool nullAware, bool nullAsValue,
               const std::vector<FieldAccessTypedExprPtr>& leftKeys, const std::vector<FieldAccessTypedExprPtr>& rightKeys,
               TypedExprPtr filter, PlanNodePtr left, PlanNodePtr right, RowTypePtr outputType)
      : PlanNode(id),
        joinType_(joinType),
        nullAware_(nullAware),
        nullAsValue_(nullAsValue),
        leftKeys_(leftKeys),
        rightKeys_(rightKeys),
        filter_(std::move(filter)),
        sources_{std::move(left), std::move(right)},
        outputType_(std::move(outputType)) {}
  JoinType joinType() const {
    return joinType_;
  }
  bool isNullAware() const {
    return nullAware_;
  }
  bool isNullAsValue() const {
    return nullAsValue_;
  }
  const std::vector<FieldAccessTypedExprPtr>& leftKeys() const {
    return leftKeys_;
  }
  const std::vector<FieldAccessTypedExprPtr>& rightKeys() const {
    return rightKeys_;
  }
  const TypedExprPtr& filter() const {
    return filter_;
  }
  bool isLeftSemiProjectJoin() const {
    return joinType_ == JoinType::kLeftSemiProject;
  }
  bool isRightSemiProjectJoin() const {
    return joinType_ == JoinType::kRightSemiProject;
  }
  bool isLeftSemiFilterJoin() const {
    return joinType_ == JoinType::kLeftSemiFilter;
  }
  bool isAntiJoin() const {
    return joinType_ == JoinType::kAnti;
  }
  // core/PlanNode.h:3391-3398
  bool canDropDuplicates() const {
    return filter_ == nullptr && (isLeftSemiFilterJoin() || isLeftSemiProjectJoin() || isAntiJoin());
  }
  const RowTypePtr& outputType() const override {
    return outputType_;
  }
  const std::vector<PlanNodePtr>& sources() const override {
    return sources_;
  }
  std::string_view name() const override {
    return "HashJoin";
  }

 private:
  const JoinType joinType_;
  const bool nullAware_, nullAsValue_;
  const std::vector<FieldAccessTypedExprPtr> leftKeys_, rightKeys_;
  const TypedExprPtr filter_;
  const std::vector<PlanNodePtr> sources_;
  const RowTypePtr outputType_;
};

struct PlanFragment {
  PlanNodePtr planNode;
};

// ---- core/QueryConfig.h ------------------------------------------------------------------------

class QueryConfig {
 public:
  uint64_t maxPartialAggregationMemoryUsage() const {
    return maxPartialAggregationMemory;
  }
  uint64_t preferredOutputBatchBytes() const {
    return preferredBatchBytes;
  }
  vector_size_t preferredOutputBatchRows() const {
    return preferredBatchRows;
  }
  int32_t abandonPartialAggregationMinRows() const {
    return 100'000;
  }
  int32_t abandonPartialAggregationMinPct() const {
    return 80;
  }
  bool hashProbeDynamicFilterPushdownEnabled() const {  // core/QueryConfig.h
    return dynamicFilterPushdown;
  }
  bool dynamicFilterPushdown{true};
  uint64_t maxPartialAggregationMemory{1L << 24};
  uint64_t preferredBatchBytes{10UL << 20};
  vector_size_t preferredBatchRows{1024};
};

}  // namespace core

// ---- type/Filter.h, common/base/SplitBlockBloomFilter.h, common/base/RuntimeMetrics.h -------------
// What HashProbe::pushdownDynamicFilters (exec/HashProbe.cpp:408-457) hands to a scan: declarations with
// the reference's names and signatures, the minimum of behaviour a test needs to look inside.

struct RuntimeCounter {  // common/base/RuntimeMetrics.h:35-42
  enum class Unit { kNone, kNanos, kBytes };
  int64_t value;
  Unit unit{Unit::kNone};
  explicit RuntimeCounter(int64_t _value, Unit _unit = Unit::kNone) : value(_value), unit(_unit) {}
};

struct RuntimeMetric {  // common/base/RuntimeMetrics.h:44-70
  RuntimeCounter::Unit unit;
  int64_t sum{0};
  uint64_t count{0};
  int64_t min{std::numeric_limits<int64_t>::max()};
  int64_t max{std::numeric_limits<int64_t>::min()};
  explicit RuntimeMetric(RuntimeCounter::Unit _unit = RuntimeCounter::Unit::kNone) : unit(_unit) {}
  explicit RuntimeMetric(int64_t value, RuntimeCounter::Unit _unit = RuntimeCounter::Unit::kNone)
      : unit(_unit), sum{value}, count{1}, min{value}, max{value} {}
  void addValue(int64_t value) {
    sum += value;
    ++count;
    min = std::min(min, value);
    max = std::max(max, value);
  }
};

struct SplitBlockBloomFilter {  // common/base/SplitBlockBloomFilter.h: a block is one xsimd::batch<uint32_t>
  struct Block {
    uint32_t words[8];
  };
  static int64_t numBlocks(int64_t numElements, double falsePositive);  // (below: the library's restatement)
};
extern "C" int64_t vx355_bloom_num_blocks(int64_t num_elements, double false_positive, int32_t lanes);
inline int64_t SplitBlockBloomFilter::numBlocks(int64_t numElements, double falsePositive) {
  return vx355_bloom_num_blocks(numElements, falsePositive, sizeof(Block) / sizeof(uint32_t));
}

namespace common {

enum class FilterKind { kBigintRange, kBigintValuesUsingHashTable, kBigintValuesUsingBitmask, kBigintValuesUsingBloomFilter, kBigintMultiRange };

class Filter {  // type/Filter.h
 public:
  Filter(bool deterministic, bool nullAllowed, FilterKind kind) : nullAllowed_(nullAllowed), deterministic_(deterministic), kind_(kind) {}
  virtual ~Filter() = default;
  FilterKind kind() const {
    return kind_;
  }
  bool nullAllowed() const {
    return nullAllowed_;
  }
  virtual bool testInt64(int64_t value) const = 0;
  /// Filter::merge (type/Filter.cpp): 'filter' AND whatever 'target' already holds.
  static void merge(const std::shared_ptr<Filter>& filter, std::shared_ptr<Filter>& target);

 protected:
  const bool nullAllowed_;
  const bool deterministic_;
  const FilterKind kind_;
};
using FilterPtr = std::shared_ptr<Filter>;

class BigintRange final : public Filter {
 public:
  BigintRange(int64_t lower, int64_t upper, bool nullAllowed)
      : Filter(true, nullAllowed, FilterKind::kBigintRange), lower_(lower), upper_(upper) {}
  bool testInt64(int64_t value) const final {
    return value >= lower_ && value <= upper_;
  }
  int64_t lower() const {
    return lower_;
  }
  int64_t upper() const {
    return upper_;
  }

 private:
  const int64_t lower_, upper_;
};

class BigintValuesUsingHashTable final : public Filter {
 public:
  BigintValuesUsingHashTable(int64_t min, int64_t max, const std::vector<int64_t>& values, bool nullAllowed)
      : Filter(true, nullAllowed, FilterKind::kBigintValuesUsingHashTable), min_(min), max_(max), values_(values) {
    std::sort(values_.begin(), values_.end());
  }
  bool testInt64(int64_t value) const final {
    return std::binary_search(values_.begin(), values_.end(), value);
  }
  int64_t min() const {
    return min_;
  }
  int64_t max() const {
    return max_;
  }
  const std::vector<int64_t>& values() const {
    return values_;
  }

 private:
  const int64_t min_, max_;
  std::vector<int64_t> values_;
};

/// type/Filter.cpp:1052-1114: a BigintRange when the values are consecutive, else a value-set filter
/// (the reference picks bitmask or hash table by density; the stand-in keeps one class).
inline std::unique_ptr<Filter> createBigintValues(const std::vector<int64_t>& values, bool nullAllowed) {
  VELOX_CHECK(!values.empty(), "createBigintValues: no values");
  const auto [lo, hi] = std::minmax_element(values.begin(), values.end());
  if (static_cast<uint64_t>(*hi) - static_cast<uint64_t>(*lo) + 1 == values.size()) {
    return std::make_unique<BigintRange>(*lo, *hi, nullAllowed);
  }
  return std::make_unique<BigintValuesUsingHashTable>(*lo, *hi, values, nullAllowed);
}

class BigintValuesUsingBloomFilter final : public Filter {  // type/Filter.h:1294-1360
 public:
  static int64_t numBlocks(int64_t capacity) {
    return SplitBlockBloomFilter::numBlocks(capacity, 0.01);
  }
  BigintValuesUsingBloomFilter(int64_t capacity, bool nullAllowed)
      : Filter(true, nullAllowed, FilterKind::kBigintValuesUsingBloomFilter), blocks_(numBlocks(capacity)) {}
  bool testInt64(int64_t /*value*/) const final {
    return true;  // (the test checks the blocks through vx355_bloom_test instead)
  }
  int64_t blocksByteSize() const {
    return static_cast<int64_t>(blocks_.size() * sizeof(SplitBlockBloomFilter::Block));
  }
  /// NOT in the reference: the accessor INTEGRATION.md section 3 asks a maintainer to add (two lines) so that
  /// blocks computed elsewhere can be installed; upstream only offers insert(value).
  SplitBlockBloomFilter::Block* mutableBlocks() {
    return blocks_.data();
  }
  int64_t numBlocksHeld() const {
    return static_cast<int64_t>(blocks_.size());
  }

 private:
  std::vector<SplitBlockBloomFilter::Block> blocks_;
};

/// (stand-in for the merged filter types of the reference: both must pass)
class AndOfFilters final : public Filter {
 public:
  AndOfFilters(FilterPtr a, FilterPtr b) : Filter(true, false, FilterKind::kBigintMultiRange), a_(std::move(a)), b_(std::move(b)) {}
  bool testInt64(int64_t value) const final {
    return a_->testInt64(value) && b_->testInt64(value);
  }

 private:
  FilterPtr a_, b_;
};

inline void Filter::merge(const FilterPtr& filter, FilterPtr& target) {
  target = target ? std::make_shared<AndOfFilters>(target, filter) : filter;
}

}  // namespace common

// ---- exec/ -------------------------------------------------------------------------------------

namespace exec {

enum class BlockingReason {
  kNotBlocked,
  kWaitForConsumer,
  kWaitForSplit,
  kWaitForProducer,
  kWaitForJoinBuild,
  kWaitForJoinProbe,
  kWaitForMergeJoinRightSide,
  kWaitForMemory,
  kWaitForConnector,
  kYield,
  kWaitForArbitration,
};

class Driver;
class Operator;
class Task;

struct DriverCtx {
  const int driverId;
  const int pipelineId;
  const uint32_t splitGroupId;
  const uint32_t partitionId;
  std::shared_ptr<Task> task;
  Driver* driver{nullptr};
  DriverCtx(std::shared_ptr<Task> _task, int _driverId, int _pipelineId, uint32_t _splitGroupId, uint32_t _partitionId)
      : driverId(_driverId), pipelineId(_pipelineId), splitGroupId(_splitGroupId), partitionId(_partitionId), task(std::move(_task)) {}
  const core::QueryConfig& queryConfig() const;
};

class OperatorCtx {
 public:
  OperatorCtx(DriverCtx* driverCtx, core::PlanNodeId planNodeId, int32_t operatorId, std::string operatorType)
      : driverCtx_(driverCtx), planNodeId_(std::move(planNodeId)), operatorId_(operatorId), operatorType_(std::move(operatorType)) {}
  const std::shared_ptr<Task>& task() const {
    return driverCtx_->task;
  }
  const std::string& taskId() const;
  Driver* driver() const {
    return driverCtx_->driver;
  }
  DriverCtx* driverCtx() const {
    return driverCtx_;
  }
  const core::PlanNodeId& planNodeId() const {
    return planNodeId_;
  }
  int32_t operatorId() const {
    return operatorId_;
  }
  void setOperatorIdFromAdapter(int32_t id) {
    operatorId_ = id;
  }
  const std::string& operatorType() const {
    return operatorType_;
  }
  memory::MemoryPool* pool() const {
    static memory::MemoryPool pool;
    return &pool;
  }

 private:
  DriverCtx* driverCtx_;
  core::PlanNodeId planNodeId_;
  int32_t operatorId_;
  std::string operatorType_;
};

struct IdentityProjection {
  column_index_t inputChannel;
  column_index_t outputChannel;
};

/// exec/Driver.h:344-355 (folly containers replaced by std ones).
struct PushdownFilters {
  std::unordered_map<column_index_t, common::FilterPtr> filters;
  std::unordered_set<column_index_t> dynamicFilteredColumns;
  bool staticFiltersInitialized = false;
};

/// exec/OperatorStats.h, the part the shim writes: runtime stats by name.
struct OperatorStats {
  std::unordered_map<std::string, RuntimeMetric> runtimeStats;
  void addRuntimeStat(std::string_view name, const RuntimeCounter& value) {
    auto it = runtimeStats.find(std::string(name));
    if (it == runtimeStats.end()) {
      runtimeStats.emplace(std::string(name), RuntimeMetric(value.value, value.unit));
    } else {
      it->second.addValue(value.value);
    }
  }
};

struct DriverStats {  // exec/Driver.h
  static constexpr std::string_view kDynamicFiltersProduced{"dynamicFiltersProduced"};
  static constexpr std::string_view kDynamicFiltersAccepted{"dynamicFiltersAccepted"};
};

class Operator {
 public:
  Operator(DriverCtx* driverCtx, RowTypePtr outputType, int32_t operatorId, std::string planNodeId, std::string_view operatorType)
      : operatorCtx_(std::make_unique<OperatorCtx>(driverCtx, std::move(planNodeId), operatorId, std::string(operatorType))),
        outputType_(std::move(outputType)) {}
  virtual ~Operator() = default;
  virtual void initialize() {
    initialized_ = true;
  }
  virtual bool needsInput() const = 0;
  virtual void addInput(RowVectorPtr input) = 0;
  virtual void noMoreInput() {
    noMoreInput_ = true;
  }
  virtual RowVectorPtr getOutput() = 0;
  virtual BlockingReason isBlocked(ContinueFuture* future) = 0;
  virtual bool isFinished() = 0;
  virtual void close() {
    input_ = nullptr;
  }
  virtual bool isFilter() const {
    return false;
  }
  virtual bool canReclaim() const {
    return false;
  }
  /// exec/Operator.h:315-329.
  virtual bool canAddDynamicFilter() const {
    return false;
  }
  virtual void addDynamicFilterLocked(const core::PlanNodeId& /*producer*/, const PushdownFilters& /*filters*/) {
    VELOX_UNSUPPORTED("This operator doesn't support dynamic filter pushdown");
  }
  /// exec/Operator.h:336.
  const std::vector<IdentityProjection>& identityProjections() const {
    return identityProjections_;
  }
  /// exec/Operator.h:359-362 (folly::Synchronized<OperatorStats> replaced by a mutex).
  void addRuntimeStat(std::string_view name, const RuntimeCounter& value) {
    std::lock_guard<std::mutex> lock(statsMutex_);
    stats_.addRuntimeStat(name, value);
  }
  /// (stub only) a copy of the stats for a test to look at
  OperatorStats statsCopy() const {
    std::lock_guard<std::mutex> lock(statsMutex_);
    return stats_;
  }
  memory::MemoryPool* pool() const {
    return operatorCtx_->pool();
  }
  const core::PlanNodeId& planNodeId() const {
    return operatorCtx_->planNodeId();
  }
  int32_t operatorId() const {
    return operatorCtx_->operatorId();
  }
  void setOperatorIdFromAdapter(int32_t id) {
    operatorCtx_->setOperatorIdFromAdapter(id);
  }
  const std::string& operatorType() const {
    return operatorCtx_->operatorType();
  }
  const OperatorCtx* operatorCtx() const {
    return operatorCtx_.get();
  }
  const RowTypePtr& outputType() const {
    return outputType_;
  }

 protected:
  vector_size_t outputBatchRows(std::optional<uint64_t> /*averageRowSize*/ = std::nullopt) const {
    return operatorCtx_->driverCtx()->queryConfig().preferredOutputBatchRows();
  }
  const std::unique_ptr<OperatorCtx> operatorCtx_;
  const RowTypePtr outputType_;
  bool initialized_{false};
  RowVectorPtr input_;
  bool noMoreInput_{false};
  std::vector<IdentityProjection> identityProjections_;
  mutable std::mutex statsMutex_;
  OperatorStats stats_;
};

using OperatorSupplier = std::function<std::unique_ptr<Operator>(int32_t operatorId, DriverCtx* ctx)>;

/// The CPU operators the adapter looks for. Here they only carry their identity: a Driver that still
/// holds one after adaptation "stayed on the CPU" (the stub cannot run it).
class CpuOperatorStandIn : public Operator {
 public:
  using Operator::Operator;
  bool needsInput() const override {
    return false;
  }
  void addInput(RowVectorPtr) override {
    VELOX_NYI("the stub has no CPU operators");
  }
  RowVectorPtr getOutput() override {
    VELOX_NYI("the stub has no CPU operators");
  }
  BlockingReason isBlocked(ContinueFuture*) override {
    return BlockingReason::kNotBlocked;
  }
  bool isFinished() override {
    return true;
  }
};

/// exec/HashTable.h:155-163: the names of the runtime stats of operators that own a hash table.
struct BaseHashTable {
  static constexpr std::string_view kCapacity{"hashtable.capacity"};
  static constexpr std::string_view kNumRehashes{"hashtable.numRehashes"};
  static constexpr std::string_view kNumDistinct{"hashtable.numDistinct"};
  static constexpr std::string_view kNumTombstones{"hashtable.numTombstones"};
  static constexpr std::string_view kHashMode{"hashtable.hashMode"};
  static constexpr std::string_view kBuildWallNanos{"hashtable.buildWallNanos"};
};

class HashAggregation : public CpuOperatorStandIn {
 public:
  static constexpr std::string_view kFlushRowCount{"flushRowCount"};  // exec/HashAggregation.h:30-38
  static constexpr std::string_view kFlushTimes{"flushTimes"};
  HashAggregation(int32_t operatorId, DriverCtx* driverCtx, const std::shared_ptr<const core::AggregationNode>& node)
      : CpuOperatorStandIn(driverCtx, node->outputType(), operatorId, node->id(), "Aggregation") {}
};

class FilterProject : public CpuOperatorStandIn {
 public:
  // exec/FilterProject.cpp: the operator carries the project node's id when there is one, else the filter's
  FilterProject(int32_t operatorId, DriverCtx* driverCtx, const std::shared_ptr<const core::FilterNode>& filter,
                const std::shared_ptr<const core::ProjectNode>& project)
      : CpuOperatorStandIn(driverCtx, project ? project->outputType() : filter->outputType(), operatorId,
                           project ? project->id() : filter->id(), "FilterProject") {}
  bool isFilter() const override {
    return true;
  }
};

class HashBuild : public CpuOperatorStandIn {
 public:
  HashBuild(int32_t operatorId, DriverCtx* driverCtx, const std::shared_ptr<const core::HashJoinNode>& node)
      : CpuOperatorStandIn(driverCtx, nullptr, operatorId, node->id(), "HashBuild") {}
};

class HashProbe : public CpuOperatorStandIn {
 public:
  HashProbe(int32_t operatorId, DriverCtx* driverCtx, const std::shared_ptr<const core::HashJoinNode>& node)
      : CpuOperatorStandIn(driverCtx, node->outputType(), operatorId, node->id(), "HashProbe") {}
};

class Driver : public std::enable_shared_from_this<Driver> {
 public:
  explicit Driver(std::unique_ptr<DriverCtx> ctx) : ctx_(std::move(ctx)) {
    ctx_->driver = this;
  }
  DriverCtx* driverCtx() const {
    return ctx_.get();
  }
  std::vector<Operator*> operators() const {
    std::vector<Operator*> out;
    for (const auto& op : operators_) {
      out.push_back(op.get());
    }
    return out;
  }
  Operator* findOperator(std::string_view planNodeId) const {
    for (const auto& op : operators_) {
      if (op->planNodeId() == planNodeId) {
        return op.get();
      }
    }
    return nullptr;
  }
  Operator* findOperator(int32_t operatorId) const {
    return operators_.at(operatorId).get();
  }
  bool shouldYield() const {
    return false;
  }
  int operatorIndex(const Operator* op) const {
    for (size_t i = 0; i < operators_.size(); ++i) {
      if (operators_[i].get() == op) {
        return static_cast<int>(i);
      }
    }
    VELOX_FAIL("operator not in this driver");
  }
  /// exec/Driver.h:463-467, exec/Driver.cpp:1183-1250: for every channel of 'filterSource', walk upstream
  /// through operators that project the channel as it is; the operator the walk stops at takes the filter
  /// if it can (a scan). Returns the number of filters produced.
  int pushdownFilters(Operator* filterSource, const std::vector<column_index_t>& channels,
                      const std::function<bool(column_index_t, common::FilterPtr&)>& makeFilter) {
    const int sourceIndex = operatorIndex(filterSource);
    pushdown_.resize(operators_.size());
    std::vector<int> accepted(sourceIndex, 0);
    int produced = 0;
    for (size_t i = 0; i < channels.size(); ++i) {
      column_index_t channel = channels[i];
      int j = sourceIndex - 1;
      for (; j > 0; --j) {
        bool identity = false;
        for (const auto& projection : operators_[j]->identityProjections()) {
          if (projection.outputChannel == channel) {
            channel = projection.inputChannel;
            identity = true;
            break;
          }
        }
        if (!identity) {
          break;
        }
      }
      if (j < 0 || !operators_[j]->canAddDynamicFilter()) {
        continue;
      }
      common::FilterPtr filter;
      if (makeFilter(static_cast<column_index_t>(i), filter)) {
        if (filter) {
          common::Filter::merge(filter, pushdown_[j].filters[channel]);
          pushdown_[j].dynamicFilteredColumns.insert(channel);
        }
        ++produced;
        ++accepted[j];
      }
    }
    for (int j = 0; j < sourceIndex; ++j) {
      if (accepted[j] > 0) {
        operators_[j]->addDynamicFilterLocked(filterSource->planNodeId(), pushdown_[j]);
        operators_[j]->addRuntimeStat(DriverStats::kDynamicFiltersAccepted, RuntimeCounter(accepted[j]));
      }
    }
    if (produced > 0) {
      filterSource->addRuntimeStat(DriverStats::kDynamicFiltersProduced, RuntimeCounter(produced));
    }
    return produced;
  }
  // (stub only) the pipeline under construction
  std::vector<std::unique_ptr<Operator>>& mutableOperators() {
    return operators_;
  }

 private:
  std::unique_ptr<DriverCtx> ctx_;
  std::vector<std::unique_ptr<Operator>> operators_;
  std::vector<PushdownFilters> pushdown_;
  friend struct DriverFactory;
};

struct DriverFactory;
using AdaptDriverFunction = std::function<bool(const DriverFactory& factory, Driver& driver)>;

struct DriverAdapter {
  std::string label;
  std::function<void(const core::PlanFragment&)> inspect;
  AdaptDriverFunction adapt;
};

struct DriverFactory {
  std::vector<std::shared_ptr<const core::PlanNode>> planNodes;
  OperatorSupplier operatorSupplier;
  uint32_t maxDrivers{1};
  uint32_t numDrivers{1};
  uint32_t numTotalDrivers{1};
  std::shared_ptr<const core::PlanNode> consumerNode;

  std::vector<std::unique_ptr<Operator>> replaceOperators(Driver& driver, int32_t begin, int32_t end,
                                                          std::vector<std::unique_ptr<Operator>> replaceWith) const {
    auto& ops = driver.operators_;
    std::vector<std::unique_ptr<Operator>> replaced;
    for (int32_t i = begin; i < end; ++i) {
      replaced.push_back(std::move(ops[i]));
    }
    ops.erase(ops.begin() + begin, ops.begin() + end);
    ops.insert(ops.begin() + begin, std::make_move_iterator(replaceWith.begin()), std::make_move_iterator(replaceWith.end()));
    for (int32_t i = 0; i < static_cast<int32_t>(ops.size()); ++i) {
      ops[i]->setOperatorIdFromAdapter(i);
    }
    return replaced;
  }
  static std::vector<DriverAdapter>& adapters() {
    static std::vector<DriverAdapter> list;
    return list;
  }
  static void registerAdapter(DriverAdapter adapter) {
    adapters().push_back(std::move(adapter));
  }
};

class Task {
 public:
  explicit Task(std::string taskId) : taskId_(std::move(taskId)) {}
  const std::string& taskId() const {
    return taskId_;
  }
  const core::QueryConfig& queryConfig() const {
    return config_;
  }
  core::QueryConfig& mutableQueryConfig() {
    return config_;
  }
  /// exec/Task.h:588-593. (stub) 'peersOf' says how many Drivers run the plan node; the last caller gets
  /// true, the promises of the earlier ones and the earlier Drivers themselves.
  bool allPeersFinished(const core::PlanNodeId& planNodeId, Driver* caller, ContinueFuture* future,
                        std::vector<ContinuePromise>& promises, std::vector<std::shared_ptr<Driver>>& peers) {
    std::lock_guard<std::mutex> l(mutex_);
    auto& state = barriers_[planNodeId];
    const int32_t expected = peersOf.count(planNodeId) ? peersOf[planNodeId] : 1;
    if (static_cast<int32_t>(state.drivers.size()) + 1 < expected) {
      state.drivers.push_back(caller->shared_from_this());
      if (future != nullptr) {
        state.promises.emplace_back("Task::allPeersFinished");
        *future = state.promises.back().getSemiFuture();
      }
      return false;
    }
    peers = std::move(state.drivers);
    promises = std::move(state.promises);
    barriers_.erase(planNodeId);
    return true;
  }
  std::map<core::PlanNodeId, int32_t> peersOf;

 private:
  struct Barrier {
    std::vector<std::shared_ptr<Driver>> drivers;
    std::vector<ContinuePromise> promises;
  };
  const std::string taskId_;
  core::QueryConfig config_;
  std::mutex mutex_;
  std::map<core::PlanNodeId, Barrier> barriers_;
};

inline const core::QueryConfig& DriverCtx::queryConfig() const {
  return task->queryConfig();
}
inline const std::string& OperatorCtx::taskId() const {
  return driverCtx_->task->taskId();
}

/// exec/Operator.h:709
inline column_index_t exprToChannel(const core::ITypedExpr* expr, const TypePtr& type) {
  if (auto field = dynamic_cast<const core::FieldAccessTypedExpr*>(expr)) {
    return type->asRow().getChildIdx(field->name());
  }
  if (dynamic_cast<const core::ConstantTypedExpr*>(expr)) {
    return kConstantChannel;
  }
  VELOX_FAIL("Expression must be field access or constant: {}", expr->toString());
}

/// exec/Aggregate.h
inline bool isRawInput(core::AggregationNode::Step step) {
  return step == core::AggregationNode::Step::kPartial || step == core::AggregationNode::Step::kSingle;
}
inline bool isPartialOutput(core::AggregationNode::Step step) {
  return step == core::AggregationNode::Step::kPartial || step == core::AggregationNode::Step::kIntermediate;
}

/// exec/OperatorUtils.h:71-75
inline VectorPtr wrapChild(vector_size_t size, BufferPtr mapping, const VectorPtr& child, BufferPtr nulls = nullptr) {
  if (mapping == nullptr) {
    return child;
  }
  return BaseVector::wrapInDictionary(std::move(nulls), std::move(mapping), size, child);
}

}  // namespace exec
}  // namespace facebook::velox
