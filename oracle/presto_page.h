// TEST INFRASTRUCTURE (see oracle/README): CPU restatement of the PrestoPage writer, the checker
// for vx355_presto_serialize. Follows the reference's iterative serializer row by row:
//   serializers/VectorStream.h:88-109      appendNull / appendNonNull / appendLength
//   serializers/VectorStream.cpp:207-299   VectorStream::flush (column layout), flushNulls
//   serializers/VectorStream.cpp:339-363   Timestamp / bool values
//   serializers/PrestoSerializerSerializationUtils.h:37-45,167-268  page header, checksum
//   serializers/PrestoSerializerSerializationUtils.cpp:997-1040    encoding names
//   common/memory/ByteStream.cpp:288-300   null bits leave the stream bit-reversed per byte
// The reference's tests hold no golden bytes for this format (PrestoSerializerTest.cpp round-trips
// through its own reader); the restatement is pinned on hand-assembled pages that follow the
// published format (prestodb.io/docs/current/develop/serialized-page.html) and on an independent
// reader in tests/presto_page_reader.py. zlib's crc32 (== folly::crc32 / boost crc_32_type)
// pins the checksum.
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "hashing.h"

namespace orc {

inline const char* prestoEncodingName(int32_t kind) {
  switch (kind) {
    case VX355_BOOLEAN:
    case VX355_TINYINT:
      return "BYTE_ARRAY";
    case VX355_SMALLINT:
      return "SHORT_ARRAY";
    case VX355_INTEGER:
    case VX355_REAL:
      return "INT_ARRAY";
    case VX355_BIGINT:
    case VX355_DOUBLE:
    case VX355_TIMESTAMP:
      return "LONG_ARRAY";
    case VX355_VARCHAR:
    case VX355_VARBINARY:
      return "VARIABLE_WIDTH";
    default:
      throw std::runtime_error("PrestoPage: unsupported type kind " + std::to_string(kind));
  }
}

// boost::crc_32_type as common/base/Crc.h drives folly::crc32: bitwise, no tables on purpose.
inline uint32_t crc32Update(uint32_t state, const unsigned char* p, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    state ^= p[i];
    for (int k = 0; k < 8; ++k) {
      state = (state & 1) ? 0xEDB88320u ^ (state >> 1) : state >> 1;
    }
  }
  return state;
}

struct UserErrorTag : std::runtime_error {
  using std::runtime_error::runtime_error;
};

class PrestoColumnStream {
 public:
  PrestoColumnStream(int32_t kind, bool lossless) : kind_(kind), lossless_(lossless) {}

  void appendNull() {
    nullBits_.push_back(true);
    ++nullCount_;
    if (isString()) {
      offsets_.push_back(totalLength_);
    }
  }

  void appendValue(const Decoded& d, int32_t row) {
    nullBits_.push_back(false);
    ++nonNullCount_;
    uint8_t tmp;
    const void* p = d.valuePtr(row, &tmp);
    switch (kind_) {
      case VX355_BOOLEAN:
        values_.push_back(tmp ? 1 : 0);
        break;
      case VX355_TIMESTAMP: {
        Timestamp ts;
        std::memcpy(&ts, p, 16);
        if (lossless_) {
          put(&ts.seconds, 8);
          put(&ts.nanos, 8);
        } else {
          // Timestamp::toMillis (type/Timestamp.h:157-172)
          __int128_t ms = static_cast<__int128_t>(ts.seconds) * 1000 + static_cast<int64_t>(ts.nanos / 1000000);
          if (ms < INT64_MIN || ms > INT64_MAX) {
            throw UserErrorTag("Could not convert Timestamp to milliseconds");
          }
          int64_t v = static_cast<int64_t>(ms);
          put(&v, 8);
        }
        break;
      }
      case VX355_VARCHAR:
      case VX355_VARBINARY: {
        const auto* sv = static_cast<const StringView*>(p);
        totalLength_ += static_cast<int32_t>(sv->size);
        offsets_.push_back(totalLength_);
        put(sv->data(), sv->size);
        break;
      }
      default:
        put(p, d.width);
    }
  }

  // VectorStream::flush
  void flush(std::vector<unsigned char>& out) const {
    const std::string name = prestoEncodingName(kind_);
    putI32(out, static_cast<int32_t>(name.size()));
    out.insert(out.end(), name.begin(), name.end());
    putI32(out, nullCount_ + nonNullCount_);
    if (isString()) {
      for (int32_t o : offsets_) {
        putI32(out, o);
      }
    }
    // flushNulls
    if (nullCount_ == 0) {
      out.push_back(0);
    } else {
      out.push_back(1);
      const size_t n = nullBits_.size();
      for (size_t b = 0; b < (n + 7) / 8; ++b) {
        unsigned char byte = 0;
        for (int j = 0; j < 8 && b * 8 + j < n; ++j) {
          if (nullBits_[b * 8 + j]) {
            byte |= static_cast<unsigned char>(0x80u >> j);  // bits::reverseBits of an LSB-first byte
          }
        }
        out.push_back(byte);
      }
    }
    if (isString()) {
      putI32(out, static_cast<int32_t>(values_.size()));
    }
    out.insert(out.end(), values_.begin(), values_.end());
  }

  static void putI32(std::vector<unsigned char>& out, int32_t v) {
    unsigned char b[4];
    std::memcpy(b, &v, 4);
    out.insert(out.end(), b, b + 4);
  }

 private:
  bool isString() const { return kind_ == VX355_VARCHAR || kind_ == VX355_VARBINARY; }
  void put(const void* p, size_t n) {
    const auto* b = static_cast<const unsigned char*>(p);
    values_.insert(values_.end(), b, b + n);
  }

  int32_t kind_;
  bool lossless_;
  int32_t nullCount_ = 0, nonNullCount_ = 0, totalLength_ = 0;
  std::vector<bool> nullBits_;
  std::vector<int32_t> offsets_;
  std::vector<unsigned char> values_;
};

// A ROW column (VX355_ROW: values = the children, nulls = the struct's own bitmap), the way
// serializeRowVector feeds a ROW VectorStream (PrestoSerializerSerializationUtils.cpp:883-919): a
// null struct appends a null and repeats the running length, a non-null struct appends length 1
// and hands its row to every child stream - the children only hold the rows of non-null structs.
// VectorStream::flush, ROW branch without nullsFirst (VectorStream.cpp:236-262): "ROW", the number
// of children, the child streams, the row count, rows + 1 offsets (the first is 0:
// VectorStream::clear, :317-324), the null flag and bits.
class PrestoRowStream {
 public:
  PrestoRowStream(const vx355_column& col, bool lossless) : nulls_(col.nulls) {
    if (col.encoding != VX355_FLAT) {
      throw std::runtime_error("PrestoPage: a ROW column must be FLAT");
    }
    const auto* kids = static_cast<const vx355_column*>(col.values);
    for (int32_t i = 0; i < col.base_size; ++i) {
      if (kids[i].type_kind == VX355_ROW) {
        throw std::runtime_error("PrestoPage: ROW inside ROW");
      }
      children_.emplace_back(kids[i].type_kind, lossless);
      decoded_.emplace_back(&kids[i]);
    }
    offsets_.push_back(0);
  }

  void append(int32_t row) {
    const bool null = nulls_ != nullptr && !((nulls_[row >> 6] >> (row & 63)) & 1);
    nullBits_.push_back(null);
    if (null) {
      ++nullCount_;
      offsets_.push_back(total_);
      return;
    }
    offsets_.push_back(++total_);
    for (size_t c = 0; c < children_.size(); ++c) {
      if (decoded_[c].isNull(row)) {
        children_[c].appendNull();
      } else {
        children_[c].appendValue(decoded_[c], row);
      }
    }
  }

  void flush(std::vector<unsigned char>& out) const {
    PrestoColumnStream::putI32(out, 3);
    out.insert(out.end(), {'R', 'O', 'W'});
    PrestoColumnStream::putI32(out, static_cast<int32_t>(children_.size()));
    for (const auto& child : children_) {
      child.flush(out);
    }
    PrestoColumnStream::putI32(out, static_cast<int32_t>(nullBits_.size()));
    for (int32_t o : offsets_) {
      PrestoColumnStream::putI32(out, o);
    }
    if (nullCount_ == 0) {
      out.push_back(0);
    } else {
      out.push_back(1);
      const size_t n = nullBits_.size();
      for (size_t b = 0; b < (n + 7) / 8; ++b) {
        unsigned char byte = 0;
        for (int j = 0; j < 8 && b * 8 + j < n; ++j) {
          if (nullBits_[b * 8 + j]) {
            byte |= static_cast<unsigned char>(0x80u >> j);
          }
        }
        out.push_back(byte);
      }
    }
  }

 private:
  const uint64_t* nulls_;
  std::vector<PrestoColumnStream> children_;
  std::vector<Decoded> decoded_;
  std::vector<bool> nullBits_;
  std::vector<int32_t> offsets_;
  int32_t nullCount_ = 0, total_ = 0;
};

// One page: flushUncompressed (PrestoSerializerSerializationUtils.h:216-268).
inline std::vector<unsigned char> prestoPage(const vx355_batch& batch, const int32_t* rows, int64_t begin,
                                             int64_t end, bool checksum, bool lossless) {
  // per column: a scalar stream or a ROW stream (index into the respective vector)
  std::vector<PrestoColumnStream> streams;
  std::vector<Decoded> decoded;
  std::vector<PrestoRowStream> rowStreams;
  std::vector<std::pair<bool, size_t>> which;
  for (int32_t c = 0; c < batch.num_cols; ++c) {
    if (batch.cols[c].type_kind == VX355_ROW) {
      which.emplace_back(true, rowStreams.size());
      rowStreams.emplace_back(batch.cols[c], lossless);
    } else {
      which.emplace_back(false, streams.size());
      streams.emplace_back(batch.cols[c].type_kind, lossless);
      decoded.emplace_back(&batch.cols[c]);
    }
  }
  for (int64_t i = begin; i < end; ++i) {
    const int32_t row = rows ? rows[i] : static_cast<int32_t>(i);
    for (int32_t c = 0; c < batch.num_cols; ++c) {
      const size_t at = which[c].second;
      if (which[c].first) {
        rowStreams[at].append(row);
      } else if (decoded[at].isNull(row)) {
        streams[at].appendNull();
      } else {
        streams[at].appendValue(decoded[at], row);
      }
    }
  }
  std::vector<unsigned char> out;
  const int32_t numRows = static_cast<int32_t>(end - begin);
  PrestoColumnStream::putI32(out, numRows);
  const unsigned char codec = checksum ? 4 : 0;
  out.push_back(codec);
  PrestoColumnStream::putI32(out, 0);
  PrestoColumnStream::putI32(out, 0);
  out.insert(out.end(), 8, 0);
  PrestoColumnStream::putI32(out, batch.num_cols);
  for (int32_t c = 0; c < batch.num_cols; ++c) {
    if (which[c].first) {
      rowStreams[which[c].second].flush(out);
    } else {
      streams[which[c].second].flush(out);
    }
  }
  const int32_t uncompressed = static_cast<int32_t>(out.size()) - 21;
  std::memcpy(out.data() + 5, &uncompressed, 4);
  std::memcpy(out.data() + 9, &uncompressed, 4);
  if (checksum) {
    uint32_t state = ~0u;
    state = crc32Update(state, out.data() + 21, static_cast<size_t>(uncompressed));
    const int codecInt = codec;  // computeChecksum takes ints and hashes their leading bytes
    state = crc32Update(state, reinterpret_cast<const unsigned char*>(&codecInt), 1);
    state = crc32Update(state, reinterpret_cast<const unsigned char*>(&numRows), 4);
    state = crc32Update(state, reinterpret_cast<const unsigned char*>(&uncompressed), 4);
    const int64_t sum = static_cast<int64_t>(static_cast<uint32_t>(~state));
    std::memcpy(out.data() + 13, &sum, 8);
  }
  return out;
}

}  // namespace orc
