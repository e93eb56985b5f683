// TEST INFRASTRUCTURE ONLY — CPU oracle, see oracle.h.
// Hash primitives of the reference's VectorHasher path, restated.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

#include "../include/vx355.h"

namespace orc {

// folly/hash/Hash.h twang_mix64 (folly v2026.01.05.00, pinned by
// /root/reference/CMake/resolve_dependency_modules/folly/CMakeLists.txt:20;
// called through folly::hasher<int64_t> at exec/VectorHasher.cpp:80 and
// directly as mixNormalizedKey at exec/HashTable.cpp:442-444).
inline uint64_t twangMix64(uint64_t key) {
  key = (~key) + (key << 21);
  key = key ^ (key >> 24);
  key = key + (key << 3) + (key << 8);
  key = key ^ (key >> 14);
  key = key + (key << 2) + (key << 4);
  key = key ^ (key >> 28);
  key = key + (key << 31);
  return key;
}

// folly/hash/Hash.h twang_32from64 (not on the path; kept because folly's
// HashTest pins it next to the other two and it cross-checks the restatement).
inline uint32_t twang32From64(uint64_t key) {
  key = (~key) + (key << 18);
  key = key ^ (key >> 31);
  key = key * 21;
  key = key ^ (key >> 11);
  key = key + (key << 6);
  key = key ^ (key >> 22);
  return static_cast<uint32_t>(key);
}

// folly/hash/Hash.h jenkins_rev_mix32; identical text in-tree at
// experimental/wave/common/Hash.h:74-86.
inline uint32_t jenkinsRevMix32(uint32_t key) {
  key += (key << 12);
  key ^= (key >> 22);
  key += (key << 4);
  key ^= (key >> 9);
  key += (key << 10);
  key ^= (key >> 2);
  key += (key << 7);
  key += (key << 12);
  return key;
}

// common/base/BitUtil.h:775-784 bits::hashMix (== folly hash_128_to_64).
inline uint64_t hashMix(uint64_t upper, uint64_t lower) {
  const uint64_t kMul = 0x9ddfea08eb382d69ULL;
  uint64_t a = (lower ^ upper) * kMul;
  a ^= (a >> 47);
  uint64_t b = (upper ^ a) * kMul;
  b ^= (b >> 47);
  b *= kMul;
  return b;
}

constexpr uint64_t kNullHash = 1;  // common/base/BitUtil.h:52

// common/base/SimdUtil-inl.h:1387-1399 — the portable CRC32-C step
// (polynomial 0x82F63B78); the SSE4.2 path (_mm_crc32_u64) computes the same.
inline uint32_t crc32U64(uint32_t checksum, uint64_t value) {
  checksum ^= static_cast<uint32_t>(value);
  for (int i = 0; i < 32; ++i) {
    checksum = (checksum >> 1) ^ (0x82F63B78 & -(checksum & 1));
  }
  checksum ^= static_cast<uint32_t>(value >> 32);
  for (int i = 0; i < 32; ++i) {
    checksum = (checksum >> 1) ^ (0x82F63B78 & -(checksum & 1));
  }
  return checksum;
}

// common/base/BitUtil.h:800-806.
inline uint64_t loadPartialWord(const uint8_t* data, int32_t size) {
  uint64_t result = 0;
  std::memcpy(&result, data, size);
  return result;
}

// common/base/BitUtil.cpp:177-225 bits::hashBytes. Note that crc32U64 takes a
// uint32_t checksum, so the 64-bit lane states are truncated on every call.
inline uint64_t hashBytes(uint64_t seed, const char* data, size_t size) {
  auto begin = reinterpret_cast<const uint8_t*>(data);
  const uint64_t kMul = 0x9ddfea08eb382d69ULL;
  if (size < 8) {
    auto word = loadPartialWord(begin, static_cast<int32_t>(size));
    uint64_t crc = crc32U64(static_cast<uint32_t>(seed), word);
    uint64_t crc2 = crc32U64(static_cast<uint32_t>(seed), word >> 32);
    return crc | (crc2 << 32);
  }
  uint64_t a0 = seed;
  uint64_t a1 = seed << 32;
  uint64_t a2 = seed >> 16;
  int32_t toGo = static_cast<int32_t>(size);
  const uint8_t* p = begin;
  auto word = [&](int i) {
    uint64_t w;
    std::memcpy(&w, p + 8 * i, 8);
    return w;
  };
  while (toGo >= 24) {
    a0 = crc32U64(static_cast<uint32_t>(a0), word(0));
    a1 = crc32U64(static_cast<uint32_t>(a1), word(1));
    a2 = crc32U64(static_cast<uint32_t>(a2), word(2));
    p += 24;
    toGo -= 24;
  }
  if (toGo > 16) {
    a0 = crc32U64(static_cast<uint32_t>(a0), word(0));
    a1 = crc32U64(static_cast<uint32_t>(a1), word(1));
    a2 = crc32U64(static_cast<uint32_t>(a2), loadPartialWord(p + 16, toGo - 16));
  } else if (toGo > 8) {
    a0 = crc32U64(static_cast<uint32_t>(a0), word(0));
    a1 = crc32U64(static_cast<uint32_t>(a1),
                  toGo == 16 ? word(1) : loadPartialWord(p + 8, toGo - 8));
  } else if (toGo > 0) {
    a0 = crc32U64(static_cast<uint32_t>(a0), toGo == 8 ? word(0) : loadPartialWord(p, toGo));
  }
  return a0 ^ ((a1 * kMul)) ^ (a2 * kMul);
}

// external/xxhash/xxhash.h XXH32 for a 4-byte input (the only length
// localExchangeHash uses, exec/HashPartitionFunction.cpp:25-30).
inline uint32_t xxh32U32(uint32_t value, uint32_t seed) {
  const uint32_t P2 = 0x85EBCA77U, P3 = 0xC2B2AE3DU, P4 = 0x27D4EB2FU, P5 = 0x165667B1U;
  uint32_t h = seed + P5 + 4;
  h += value * P3;
  h = ((h << 17) | (h >> 15)) * P4;
  h ^= h >> 15;
  h *= P2;
  h ^= h >> 13;
  h *= P3;
  h ^= h >> 16;
  return h;
}

// common/base/BitUtil.h:379-386 bits::reverseBits: reverses the bits INSIDE
// each byte (byte order is kept).
inline uint32_t reverseBitsPerByte(uint32_t v) {
  uint32_t out = 0;
  for (int i = 0; i < 4; ++i) {
    uint8_t b = (v >> (8 * i)) & 0xff, r = 0;
    for (int k = 0; k < 8; ++k) {
      r |= ((b >> k) & 1) << (7 - k);
    }
    out |= static_cast<uint32_t>(r) << (8 * i);
  }
  return out;
}

// The 16-byte StringView of type/StringView.h:76-77.
struct StringView {
  uint32_t size;
  char prefix[4];
  union {
    char inlined[8];
    const char* data;
  } value;
  bool isInline() const { return size <= 12; }
  const char* data() const { return isInline() ? prefix : value.data; }
};
static_assert(sizeof(StringView) == 16, "StringView layout");

// type/Timestamp.h: {int64 seconds; uint64 nanos}.
struct Timestamp {
  int64_t seconds;
  uint64_t nanos;
};

// folly::hasher<integral> (folly/hash/Hash.h integral_hasher): <= 4 bytes ->
// jenkins_rev_mix32 of the value sign-extended to int32; 8 bytes -> twang_mix64.
inline uint64_t hashInt32Like(int32_t v) {
  return jenkinsRevMix32(static_cast<uint32_t>(v));
}
inline uint64_t hashInt64(int64_t v) { return twangMix64(static_cast<uint64_t>(v)); }
// folly::hasher<bool>: all output bits depend on the input.
inline uint64_t hashBool(bool b) { return b ? std::numeric_limits<uint64_t>::max() : 0; }
// folly::hasher<float|double> (float_hasher): 0.0 and -0.0 hash to 0, else
// twang_mix64 of the zero-extended bit pattern; wrapped by NaNAwareHash
// (type/FloatingPointUtil.h:100-109): every NaN hashes like quiet_NaN.
inline uint64_t hashDouble(double v) {
  if (std::isnan(v)) {
    v = std::numeric_limits<double>::quiet_NaN();
  }
  if (v == 0.0) {
    return 0;
  }
  uint64_t u = 0;
  std::memcpy(&u, &v, 8);
  return twangMix64(u);
}
inline uint64_t hashFloat(float v) {
  if (std::isnan(v)) {
    v = std::numeric_limits<float>::quiet_NaN();
  }
  if (v == 0.0f) {
    return 0;
  }
  uint64_t u = 0;
  std::memcpy(&u, &v, 4);
  return twangMix64(u);
}

// hashOne (exec/VectorHasher.cpp:61-83) for one non-null value of 'kind'.
inline uint64_t hashValue(int32_t kind, const void* p) {
  switch (kind) {
    case VX355_BOOLEAN:
      return hashBool(*static_cast<const uint8_t*>(p) != 0);
    case VX355_TINYINT:
      return hashInt32Like(*static_cast<const int8_t*>(p));
    case VX355_SMALLINT: {
      int16_t v;
      std::memcpy(&v, p, 2);
      return hashInt32Like(v);
    }
    case VX355_INTEGER: {
      int32_t v;
      std::memcpy(&v, p, 4);
      return hashInt32Like(v);
    }
    case VX355_BIGINT: {
      int64_t v;
      std::memcpy(&v, p, 8);
      return hashInt64(v);
    }
    case VX355_REAL: {
      float v;
      std::memcpy(&v, p, 4);
      return hashFloat(v);
    }
    case VX355_DOUBLE: {
      double v;
      std::memcpy(&v, p, 8);
      return hashDouble(v);
    }
    case VX355_VARCHAR:
    case VX355_VARBINARY: {
      // type/StringView.h:374-378: bits::hashBytes(1, data, size).
      auto* sv = static_cast<const StringView*>(p);
      return hashBytes(1, sv->data(), sv->size);
    }
    case VX355_TIMESTAMP: {
      // type/Timestamp.h:451-455: hashMix(seconds, nanos).
      auto* ts = static_cast<const Timestamp*>(p);
      return hashMix(static_cast<uint64_t>(ts->seconds), ts->nanos);
    }
    default:
      return 0;
  }
}

inline bool bitSet(const uint64_t* bits, int64_t i) { return (bits[i >> 6] >> (i & 63)) & 1; }
inline void setBit(uint64_t* bits, int64_t i, bool v) {
  if (v) {
    bits[i >> 6] |= (1ULL << (i & 63));
  } else {
    bits[i >> 6] &= ~(1ULL << (i & 63));
  }
}

inline int kindWidth(int32_t kind) {
  switch (kind) {
    case VX355_BOOLEAN:
      return 0;  // bit packed
    case VX355_TINYINT:
      return 1;
    case VX355_SMALLINT:
      return 2;
    case VX355_INTEGER:
    case VX355_REAL:
      return 4;
    case VX355_BIGINT:
    case VX355_DOUBLE:
      return 8;
    case VX355_VARCHAR:
    case VX355_VARBINARY:
    case VX355_TIMESTAMP:
      return 16;
    default:
      return -1;
  }
}

// DecodedVector (vector/DecodedVector.h) over a vx355_column.
struct Decoded {
  const vx355_column* c;
  int width;
  explicit Decoded(const vx355_column* col) : c(col), width(kindWidth(col->type_kind)) {}
  bool isConstant() const { return c->encoding == VX355_CONSTANT; }
  bool isNull(int32_t row) const {
    if (!c->nulls) {
      return false;
    }
    return !bitSet(c->nulls, isConstant() ? 0 : row);
  }
  int32_t index(int32_t row) const {
    switch (c->encoding) {
      case VX355_CONSTANT:
        return 0;
      case VX355_DICTIONARY:
        return c->indices[row];
      default:
        return row;
    }
  }
  // Pointer to the value bytes; for BOOLEAN fills 'tmp' with 0/1.
  const void* valuePtr(int32_t row, uint8_t* tmp) const {
    int32_t i = index(row);
    if (width == 0) {
      *tmp = bitSet(static_cast<const uint64_t*>(c->values), i) ? 1 : 0;
      return tmp;
    }
    return static_cast<const char*>(c->values) + static_cast<int64_t>(i) * width;
  }
  // Integer-like kinds widened to int64 (VectorHasher::toInt64).
  int64_t int64At(int32_t row) const {
    uint8_t tmp = 0;
    const void* p = valuePtr(row, &tmp);
    switch (c->type_kind) {
      case VX355_BOOLEAN:
        return tmp;
      case VX355_TINYINT:
        return *static_cast<const int8_t*>(p);
      case VX355_SMALLINT: {
        int16_t v;
        std::memcpy(&v, p, 2);
        return v;
      }
      case VX355_INTEGER: {
        int32_t v;
        std::memcpy(&v, p, 4);
        return v;
      }
      default: {
        int64_t v;
        std::memcpy(&v, p, 8);
        return v;
      }
    }
  }
  double doubleAt(int32_t row) const {
    uint8_t tmp = 0;
    const void* p = valuePtr(row, &tmp);
    if (c->type_kind == VX355_REAL) {
      float v;
      std::memcpy(&v, p, 4);
      return v;
    }
    if (c->type_kind == VX355_DOUBLE) {
      double v;
      std::memcpy(&v, p, 8);
      return v;
    }
    return static_cast<double>(int64At(row));
  }
  uint64_t hashAt(int32_t row) const {
    uint8_t tmp = 0;
    return hashValue(c->type_kind, valuePtr(row, &tmp));
  }
};

}  // namespace orc
