// Device-side building blocks shared by every kernel of libvx355: decoded
// column views, Velox's hash arithmetic (exec/VectorHasher.cpp:61-126,
// common/base/BitUtil.h:775-784, folly/hash/Hash.h), value-id arithmetic
// (exec/VectorHasher.h:383-387,560-566) and 64-lane wavefront primitives.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vx355.h"

namespace vx {

constexpr int kWave = 64;  // gfx950 wavefront

// What DecodedVector exposes (vector/DecodedVector.h), device resident.
struct ColView {
  const void* values = nullptr;
  const uint64_t* nulls = nullptr;   // bit per top-level row, 1 = valid
  const int32_t* indices = nullptr;  // DICTIONARY
  int32_t kind = 0;
  int32_t enc = 0;
};

struct StringView16 {
  uint32_t size;
  uint32_t prefix;
  uint64_t tail;  // 8 inline bytes (size <= 12) or a pointer
};

__host__ __device__ inline bool bitAt(const uint64_t* bits, int64_t i) {
  return (bits[i >> 6] >> (i & 63)) & 1;
}

// Workgroup barrier used everywhere instead of __syncthreads(): LDS atomics
// that return nothing (ds_add / ds_min / ds_max) are fire-and-forget, and the
// workgroup is about to read what they produce, so this wave's LDS operations
// are drained (lgkmcnt(0)) before it signals the barrier. hipcc's own
// __syncthreads() was observed to leave that wait out on a loop-exit path of
// the histogram kernels: under heavy same-bin contention the counts of one
// wave instruction went missing.
__device__ inline void blockSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0xc07f);  // vmcnt(63) expcnt(7) lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ inline int lane() { return __lane_id(); }
__device__ inline uint64_t ballot(bool p) { return __ballot(p); }
// Number of set bits of m below this lane.
__device__ inline int lanePrefix(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                   __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
}
__device__ inline int popc64(uint64_t m) { return __popcll(m); }
__device__ inline uint32_t readFirst(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline uint64_t readFirst64(uint64_t v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
  return (static_cast<uint64_t>(hi) << 32) | lo;
}
__device__ inline uint64_t shfl64(uint64_t v, int src) {
  uint32_t lo = __shfl(static_cast<uint32_t>(v), src, kWave);
  uint32_t hi = __shfl(static_cast<uint32_t>(v >> 32), src, kWave);
  return (static_cast<uint64_t>(hi) << 32) | lo;
}

// ---- folly::hasher<T> restated (oracle/hashing.h pins these against folly's
// own known-answer vectors) ----
__host__ __device__ inline uint64_t twangMix64(uint64_t key) {
  key = (~key) + (key << 21);
  key = key ^ (key >> 24);
  key = key + (key << 3) + (key << 8);
  key = key ^ (key >> 14);
  key = key + (key << 2) + (key << 4);
  key = key ^ (key >> 28);
  key = key + (key << 31);
  return key;
}
__host__ __device__ inline uint32_t jenkinsRevMix32(uint32_t key) {
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
// bits::hashMix (common/base/BitUtil.h:775-784).
__host__ __device__ inline uint64_t hashMix(uint64_t upper, uint64_t lower) {
  const uint64_t kMul = 0x9ddfea08eb382d69ULL;
  uint64_t a = (lower ^ upper) * kMul;
  a ^= (a >> 47);
  uint64_t b = (upper ^ a) * kMul;
  b ^= (b >> 47);
  b *= kMul;
  return b;
}
constexpr uint64_t kNullHash = 1;  // common/base/BitUtil.h:52

// Home slot of a VectorHasher hash in the generic-mode ({tag, row} / {tag, group}) tables. Not the
// hash's low bits: behind a hash-partitioned exchange every row of a rank has the same
// hash % numRanks (HashPartitionFunction.cpp:112-115) - for a power-of-two rank count the same low
// bits - and a table indexed by them would use one slot in numRanks. A multiplicative mix spreads
// any residue class over the whole table; the tag stays the hash's high half.
__host__ __device__ inline uint64_t slotOfHash(uint64_t hash, uint64_t mask) {
  return ((hash * 0x9E3779B97F4A7C15ULL) >> 20) & mask;
}

// CRC32-C step over 8 bytes (common/base/SimdUtil-inl.h:1387-1399), 4 bits at
// a time through a 16-entry table kept in registers/constant space.
__device__ inline uint32_t crc32cNibbles(uint32_t crc, uint32_t word) {
  // T[i] = i folded 4 times through the reflected polynomial 0x82F63B78.
  const uint32_t T[16] = {0x00000000u, 0x105EC76Fu, 0x20BD8EDEu, 0x30E349B1u, 0x417B1DBCu,
                          0x5125DAD3u, 0x61C69362u, 0x7198540Du, 0x82F63B78u, 0x92A8FC17u,
                          0xA24BB5A6u, 0xB21572C9u, 0xC38D26C4u, 0xD3D3E1ABu, 0xE330A81Au,
                          0xF36E6F75u};
  crc ^= word;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    crc = (crc >> 4) ^ T[crc & 15];
  }
  return crc;
}
__device__ inline uint32_t crc32U64(uint32_t checksum, uint64_t value) {
  checksum = crc32cNibbles(checksum, static_cast<uint32_t>(value));
  checksum = crc32cNibbles(checksum, static_cast<uint32_t>(value >> 32));
  return checksum;
}
__device__ inline uint64_t loadPartialWord(const uint8_t* p, int size) {
  uint64_t r = 0;
  for (int i = 0; i < size; ++i) {
    r |= static_cast<uint64_t>(p[i]) << (8 * i);
  }
  return r;
}
__device__ inline uint64_t loadWord(const uint8_t* p) { return loadPartialWord(p, 8); }

// bits::hashBytes (common/base/BitUtil.cpp:177-225).
__device__ inline uint64_t hashBytes(uint64_t seed, const uint8_t* data, int32_t size) {
  const uint64_t kMul = 0x9ddfea08eb382d69ULL;
  if (size < 8) {
    uint64_t word = loadPartialWord(data, size);
    uint64_t crc = crc32U64(static_cast<uint32_t>(seed), word);
    uint64_t crc2 = crc32U64(static_cast<uint32_t>(seed), word >> 32);
    return crc | (crc2 << 32);
  }
  uint64_t a0 = seed, a1 = seed << 32, a2 = seed >> 16;
  int32_t toGo = size;
  const uint8_t* p = data;
  while (toGo >= 24) {
    a0 = crc32U64(static_cast<uint32_t>(a0), loadWord(p));
    a1 = crc32U64(static_cast<uint32_t>(a1), loadWord(p + 8));
    a2 = crc32U64(static_cast<uint32_t>(a2), loadWord(p + 16));
    p += 24;
    toGo -= 24;
  }
  if (toGo > 16) {
    a0 = crc32U64(static_cast<uint32_t>(a0), loadWord(p));
    a1 = crc32U64(static_cast<uint32_t>(a1), loadWord(p + 8));
    a2 = crc32U64(static_cast<uint32_t>(a2), loadPartialWord(p + 16, toGo - 16));
  } else if (toGo > 8) {
    a0 = crc32U64(static_cast<uint32_t>(a0), loadWord(p));
    a1 = crc32U64(static_cast<uint32_t>(a1),
                  toGo == 16 ? loadWord(p + 8) : loadPartialWord(p + 8, toGo - 8));
  } else if (toGo > 0) {
    a0 = crc32U64(static_cast<uint32_t>(a0), toGo == 8 ? loadWord(p) : loadPartialWord(p, toGo));
  }
  return a0 ^ (a1 * kMul) ^ (a2 * kMul);
}

// external/xxhash XXH32 of one 4-byte value (exec/HashPartitionFunction.cpp:25-30).
__host__ __device__ inline uint32_t xxh32U32(uint32_t value, uint32_t seed) {
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
// bits::reverseBits (common/base/BitUtil.h:379-386): bit order inside each byte.
__host__ __device__ inline uint32_t reverseBitsPerByte(uint32_t v) {
  v = ((v & 0xF0F0F0F0u) >> 4) | ((v & 0x0F0F0F0Fu) << 4);
  v = ((v & 0xCCCCCCCCu) >> 2) | ((v & 0x33333333u) << 2);
  v = ((v & 0xAAAAAAAAu) >> 1) | ((v & 0x55555555u) << 1);
  return v;
}

// ---- decoded column access ----
__device__ inline bool colIsNull(const ColView& c, int64_t row) {
  if (!c.nulls) {
    return false;
  }
  return !bitAt(c.nulls, c.enc == VX355_CONSTANT ? 0 : row);
}
__device__ inline int64_t colIndex(const ColView& c, int64_t row) {
  if (c.enc == VX355_FLAT) {
    return row;
  }
  if (c.enc == VX355_DICTIONARY) {
    return c.indices[row];
  }
  return 0;
}
// Integer-like kinds widened to int64 (VectorHasher::toInt64).
__device__ inline int64_t loadInt64(const ColView& c, int64_t i) {
  switch (c.kind) {
    case VX355_BOOLEAN:
      return bitAt(static_cast<const uint64_t*>(c.values), i) ? 1 : 0;
    case VX355_TINYINT:
      return static_cast<const int8_t*>(c.values)[i];
    case VX355_SMALLINT:
      return static_cast<const int16_t*>(c.values)[i];
    case VX355_INTEGER:
      return static_cast<const int32_t*>(c.values)[i];
    default:
      return static_cast<const int64_t*>(c.values)[i];
  }
}
__device__ inline double loadDouble(const ColView& c, int64_t i) {
  if (c.kind == VX355_DOUBLE) {
    return static_cast<const double*>(c.values)[i];
  }
  if (c.kind == VX355_REAL) {
    return static_cast<double>(static_cast<const float*>(c.values)[i]);
  }
  return static_cast<double>(loadInt64(c, i));
}
__device__ inline StringView16 loadView(const ColView& c, int64_t i) {
  const uint4 raw = static_cast<const uint4*>(c.values)[i];
  StringView16 v;
  v.size = raw.x;
  v.prefix = raw.y;
  v.tail = (static_cast<uint64_t>(raw.w) << 32) | raw.z;
  return v;
}

// VectorHasher::stringAsNumber (exec/VectorHasher.h:383-387) for an inline
// view of size <= 7: the bytes as a little-endian number plus a marker bit
// above the last byte.
__device__ inline int64_t stringAsNumber(const StringView16& v) {
  uint64_t bytes = static_cast<uint64_t>(v.prefix) | (v.tail << 32);
  uint32_t size = v.size;
  uint64_t mask = size >= 8 ? ~0ULL : ((1ULL << (8 * size)) - 1);
  uint64_t n = bytes & mask;
  if (size) {
    n += 1ULL << (8 * size);
  }
  return static_cast<int64_t>(n);
}
constexpr uint32_t kStringAsRangeMaxSize = 7;  // exec/VectorHasher.h:135

// hashOne (exec/VectorHasher.cpp:61-83) of the non-null value at base index i.
__device__ inline uint64_t hashValueAt(const ColView& c, int64_t i) {
  switch (c.kind) {
    case VX355_BOOLEAN:
      return bitAt(static_cast<const uint64_t*>(c.values), i) ? ~0ULL : 0ULL;
    case VX355_TINYINT:
    case VX355_SMALLINT:
    case VX355_INTEGER:
      return jenkinsRevMix32(static_cast<uint32_t>(static_cast<int32_t>(loadInt64(c, i))));
    case VX355_BIGINT:
      return twangMix64(static_cast<uint64_t>(loadInt64(c, i)));
    case VX355_REAL: {
      float f = static_cast<const float*>(c.values)[i];
      if (f != f) {
        return twangMix64(0x7fc00000ULL);  // quiet_NaN (type/FloatingPointUtil.h:100-109)
      }
      if (f == 0.0f) {
        return 0;
      }
      return twangMix64(static_cast<uint64_t>(__float_as_uint(f)));
    }
    case VX355_DOUBLE: {
      double d = static_cast<const double*>(c.values)[i];
      if (d != d) {
        return twangMix64(0x7ff8000000000000ULL);
      }
      if (d == 0.0) {
        return 0;
      }
      return twangMix64(static_cast<uint64_t>(__double_as_longlong(d)));
    }
    case VX355_VARCHAR:
    case VX355_VARBINARY: {
      StringView16 v = loadView(c, i);
      if (v.size <= 12) {
        uint8_t buf[12];
        uint64_t lo = static_cast<uint64_t>(v.prefix) | (v.tail << 32);
        uint32_t hi = static_cast<uint32_t>(v.tail >> 32);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          buf[k] = static_cast<uint8_t>(lo >> (8 * k));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          buf[8 + k] = static_cast<uint8_t>(hi >> (8 * k));
        }
        return hashBytes(1, buf, static_cast<int32_t>(v.size));
      }
      return hashBytes(1, reinterpret_cast<const uint8_t*>(v.tail), static_cast<int32_t>(v.size));
    }
    case VX355_TIMESTAMP: {
      const uint64_t* p = static_cast<const uint64_t*>(c.values) + 2 * i;
      return hashMix(p[0], p[1]);  // type/Timestamp.h:451-455
    }
    default:
      return 0;
  }
}

// One key of a normalized key (VectorHasher in range mode, after
// enableValueRange; exec/VectorHasher.cpp:923-944).
struct KeyRange {
  int64_t min = 0;
  int64_t max = -1;
  uint64_t multiplier = 1;
  uint64_t rangeSize = 0;  // max - min + 2 (slot 0 = null); bool: 3
};

// Value id of the non-null value at base index i: value - min + 1, or 0 when
// the value cannot be mapped (exec/VectorHasher.h:560-566). 'outValue'
// receives the int64 image used for range statistics; 'mappable' is false for
// strings longer than 7 bytes.
__device__ inline uint64_t valueIdAt(const ColView& c, int64_t i, const KeyRange& r, int64_t* outValue,
                                     bool* mappable) {
  *mappable = true;
  if (c.kind == VX355_BOOLEAN) {
    int64_t b = loadInt64(c, i);
    *outValue = b;
    return b ? 2 : 1;  // exec/VectorHasher.h:771-773
  }
  int64_t v;
  if (c.kind == VX355_VARCHAR || c.kind == VX355_VARBINARY) {
    StringView16 sv = loadView(c, i);
    if (sv.size > kStringAsRangeMaxSize) {
      *mappable = false;
      *outValue = 0;
      return 0;
    }
    v = stringAsNumber(sv);
  } else {
    v = loadInt64(c, i);
  }
  *outValue = v;
  if (v < r.min || v > r.max) {
    return 0;
  }
  return static_cast<uint64_t>(v) - static_cast<uint64_t>(r.min) + 1;
}

// Comparable image of the non-null key value at base index i: integers as
// int64, floating point canonicalised (one NaN, +0.0 for both zeros: the same
// classes hashOne hashes together), strings / timestamps as two words.
__device__ inline void keyImage(const ColView& c, int64_t i, uint64_t* w0, uint64_t* w1, bool* supported) {
  *w1 = 0;
  switch (c.kind) {
    case VX355_REAL: {
      float f = static_cast<const float*>(c.values)[i];
      uint32_t b = f != f ? 0x7fc00000u : (f == 0.0f ? 0u : __float_as_uint(f));
      *w0 = b;
      break;
    }
    case VX355_DOUBLE: {
      double d = static_cast<const double*>(c.values)[i];
      *w0 = d != d ? 0x7ff8000000000000ULL
                   : (d == 0.0 ? 0ULL : static_cast<uint64_t>(__double_as_longlong(d)));
      break;
    }
    case VX355_VARCHAR:
    case VX355_VARBINARY: {
      const StringView16 v = loadView(c, i);
      if (v.size > 12) {
        *supported = false;
      }
      *w0 = static_cast<uint64_t>(v.size) | (static_cast<uint64_t>(v.prefix) << 32);
      *w1 = v.tail;
      break;
    }
    case VX355_TIMESTAMP: {
      const uint64_t* p = static_cast<const uint64_t*>(c.values) + 2 * i;
      *w0 = p[0];
      *w1 = p[1];
      break;
    }
    default:
      *w0 = static_cast<uint64_t>(loadInt64(c, i));
      break;
  }
}

// hashOne of a key from its stored image (keyImage above) and its type kind:
// the same value hashValueAt computes from the column.
__device__ inline uint64_t hashFromImage(int32_t kind, uint64_t w0, uint64_t w1) {
  switch (kind) {
    case VX355_BOOLEAN:
      return w0 ? ~0ULL : 0ULL;
    case VX355_TINYINT:
    case VX355_SMALLINT:
    case VX355_INTEGER:
      return jenkinsRevMix32(static_cast<uint32_t>(static_cast<int32_t>(static_cast<int64_t>(w0))));
    case VX355_BIGINT:
      return twangMix64(w0);
    case VX355_REAL:
    case VX355_DOUBLE:
      return w0 == 0 ? 0 : twangMix64(w0);  // the image is canonical: one NaN, +0.0 for both zeros
    case VX355_VARCHAR:
    case VX355_VARBINARY: {
      const uint32_t size = static_cast<uint32_t>(w0);
      uint8_t buf[12];
      const uint64_t lo = (w0 >> 32) | (w1 << 32);
      const uint32_t hi = static_cast<uint32_t>(w1 >> 32);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        buf[k] = static_cast<uint8_t>(lo >> (8 * k));
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        buf[8 + k] = static_cast<uint8_t>(hi >> (8 * k));
      }
      if (size > 12) {
        // non-inline: w1 is the pointer (the build side's copy in its string arena)
        return hashBytes(1, reinterpret_cast<const uint8_t*>(w1), static_cast<int32_t>(size));
      }
      return hashBytes(1, buf, static_cast<int32_t>(size));
    }
    case VX355_TIMESTAMP:
      return hashMix(w0, w1);
    default:
      return 0;
  }
}

// Equality of two string key images {size | prefix, tail-or-pointer}: inline strings by
// their words, longer ones by size, prefix and content.
__device__ inline bool stringImagesEqual(uint64_t a0, uint64_t a1, uint64_t b0, uint64_t b1) {
  if (a0 != b0) {
    return false;
  }
  const uint32_t size = static_cast<uint32_t>(a0);
  if (size <= 12) {
    return a1 == b1;
  }
  const uint8_t* pa = reinterpret_cast<const uint8_t*>(a1);
  const uint8_t* pb = reinterpret_cast<const uint8_t*>(b1);
  for (uint32_t i = 4; i < size; ++i) {  // the first 4 bytes are the prefix, already equal
    if (pa[i] != pb[i]) {
      return false;
    }
  }
  return true;
}

__device__ inline uint64_t loadAgent(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void storeAgent(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Doubles mapped to unsigned keys whose integer order is Velox's NaN-aware
// order (NaN greater than +inf; functions/lib/aggregates/MinMaxAggregateBase.cpp
// :174-184), so min/max run as integer atomics.
__host__ __device__ inline uint64_t doubleToOrdered(double d) {
  uint64_t b;
  if (d != d) {
    b = 0x7ff8000000000000ULL;
  } else {
#ifdef __HIP_DEVICE_COMPILE__
    b = static_cast<uint64_t>(__double_as_longlong(d));
#else
    __builtin_memcpy(&b, &d, 8);
#endif
  }
  return (b & 0x8000000000000000ULL) ? ~b : (b | 0x8000000000000000ULL);
}
__host__ __device__ inline double orderedToDouble(uint64_t k) {
  uint64_t b = (k & 0x8000000000000000ULL) ? (k & 0x7fffffffffffffffULL) : ~k;
  double d;
#ifdef __HIP_DEVICE_COMPILE__
  d = __longlong_as_double(static_cast<long long>(b));
#else
  __builtin_memcpy(&d, &b, 8);
#endif
  return d;
}
// int64 mapped to unsigned keys with the same order.
__host__ __device__ inline uint64_t int64ToOrdered(int64_t v) {
  return static_cast<uint64_t>(v) ^ 0x8000000000000000ULL;
}
__host__ __device__ inline int64_t orderedToInt64(uint64_t k) {
  return static_cast<int64_t>(k ^ 0x8000000000000000ULL);
}

}  // namespace vx
