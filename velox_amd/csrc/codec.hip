// Block codecs of compressed PrestoPages (host side). The reference hands the page body to a
// folly::compression::Codec chosen by PrestoOptions::compressionKind
// (common/compression/Compression.cpp:27-46, serializers/PrestoSerializer.cpp:144-145,185-199;
// writer: PrestoSerializerSerializationUtils.h:279-334). folly is not part of /root/reference; what
// its codecs put on the wire are the published formats, restated here:
//   LZ4    (folly CodecType::LZ4)    one raw LZ4 block, the uncompressed length comes from the page header
//   SNAPPY (CodecType::SNAPPY)       raw snappy: varint uncompressed length, then literal / copy elements
//   ZSTD   (CodecType::ZSTD)         a zstd frame
//   ZLIB   (CodecType::ZLIB)         RFC 1950 zlib stream
//   GZIP   (CodecType::GZIP)         RFC 1952 gzip member
// LZ4 and snappy are decoded and encoded by the code below (both formats are a page of
// specification); zstd and zlib / gzip go through the system's libzstd.so.1 / libz.so.1, loaded on
// first use with dlopen so that libvx355.so itself has no new dependency (a host without them gets
// VX355_EUNSUPPORTED for those kinds only). Pages are host memory at this boundary (they come off
// the wire), and the decoded body goes to HBM in one copy: the decode itself is host work.
#include "common.h"

#include <dlfcn.h>
#include <zlib.h>   // types and constants only: the functions are resolved at run time

#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace vx {
namespace {

// ---- LZ4 block format (lz4_Block_format.md): sequences of token | literal length+ | literals |
// offset (2 bytes LE) | match length+; the last sequence ends after its literals.
bool lz4Decode(const unsigned char* src, size_t n, unsigned char* dst, size_t dstLen) {
  const unsigned char* ip = src;
  const unsigned char* const iend = src + n;
  unsigned char* op = dst;
  unsigned char* const oend = dst + dstLen;
  while (ip < iend) {
    const unsigned token = *ip++;
    size_t lit = token >> 4;
    if (lit == 15) {
      unsigned b;
      do {
        if (ip >= iend) {
          return false;
        }
        b = *ip++;
        lit += b;
      } while (b == 255);
    }
    if (lit > static_cast<size_t>(iend - ip) || lit > static_cast<size_t>(oend - op)) {
      return false;
    }
    std::memcpy(op, ip, lit);
    op += lit;
    ip += lit;
    if (ip >= iend) {
      break;
    }
    if (iend - ip < 2) {
      return false;
    }
    const size_t offset = static_cast<size_t>(ip[0]) | (static_cast<size_t>(ip[1]) << 8);
    ip += 2;
    if (offset == 0 || offset > static_cast<size_t>(op - dst)) {
      return false;
    }
    size_t len = token & 15;
    if (len == 15) {
      unsigned b;
      do {
        if (ip >= iend) {
          return false;
        }
        b = *ip++;
        len += b;
      } while (b == 255);
    }
    len += 4;
    if (len > static_cast<size_t>(oend - op)) {
      return false;
    }
    const unsigned char* from = op - offset;
    if (offset >= len) {
      std::memcpy(op, from, len);
      op += len;
    } else {
      for (size_t i = 0; i < len; ++i) {   // overlapping: the match repeats what it is writing
        *op++ = *from++;
      }
    }
  }
  return op == oend;
}

inline uint32_t load32(const unsigned char* p) {
  uint32_t v;
  std::memcpy(&v, p, 4);
  return v;
}

void lz4PutLength(std::vector<unsigned char>& out, size_t rest) {
  while (rest >= 255) {
    out.push_back(255);
    rest -= 255;
  }
  out.push_back(static_cast<unsigned char>(rest));
}

void lz4Sequence(std::vector<unsigned char>& out, const unsigned char* lit, size_t litLen, size_t offset, size_t matchLen) {
  const size_t ml = matchLen ? matchLen - 4 : 0;
  out.push_back(static_cast<unsigned char>((litLen >= 15 ? 15 : litLen) << 4 | (ml >= 15 ? 15 : ml)));
  if (litLen >= 15) {
    lz4PutLength(out, litLen - 15);
  }
  out.insert(out.end(), lit, lit + litLen);
  if (matchLen) {
    out.push_back(static_cast<unsigned char>(offset & 0xff));
    out.push_back(static_cast<unsigned char>(offset >> 8));
    if (ml >= 15) {
      lz4PutLength(out, ml - 15);
    }
  }
}

// Greedy single-probe matcher. The format's end conditions: the last 5 bytes are literals and the
// last match starts at least 12 bytes before the end.
void lz4Encode(const unsigned char* src, size_t n, std::vector<unsigned char>& out) {
  out.clear();
  out.reserve(n / 2 + 16);
  constexpr int kHashBits = 16;
  std::vector<uint32_t> table(size_t(1) << kHashBits, 0);   // position + 1 of the last 4-byte group with this hash
  const unsigned char* anchor = src;
  size_t i = 0;
  if (n >= 13) {
    const size_t matchLimit = n - 12;   // a match may start at i <= matchLimit
    const size_t copyLimit = n - 5;     // and must end at or before here
    while (i <= matchLimit) {
      const uint32_t h = (load32(src + i) * 2654435761u) >> (32 - kHashBits);
      const size_t cand = table[h];
      table[h] = static_cast<uint32_t>(i + 1);
      if (cand && i + 1 - cand <= 65535 && load32(src + cand - 1) == load32(src + i)) {
        const size_t from = cand - 1;
        size_t len = 4;
        while (i + len < copyLimit && src[from + len] == src[i + len]) {
          ++len;
        }
        lz4Sequence(out, anchor, static_cast<size_t>(src + i - anchor), i - from, len);
        i += len;
        anchor = src + i;
      } else {
        ++i;
      }
    }
  }
  lz4Sequence(out, anchor, static_cast<size_t>(src + n - anchor), 0, 0);
}

// ---- snappy (format_description.txt): varint32 uncompressed length, then elements whose tag byte's
// low two bits say literal (00), copy with 1-byte offset (01), 2-byte offset (10), 4-byte offset (11).
bool snappyLength(const unsigned char* src, size_t n, size_t* pos, uint64_t* len) {
  uint64_t v = 0;
  for (int shift = 0; shift <= 28; shift += 7) {
    if (*pos >= n) {
      return false;
    }
    const unsigned b = src[(*pos)++];
    v |= static_cast<uint64_t>(b & 0x7f) << shift;
    if (!(b & 0x80)) {
      *len = v;
      return true;
    }
  }
  return false;
}

bool snappyDecode(const unsigned char* src, size_t n, unsigned char* dst, size_t dstLen) {
  size_t ip = 0;
  uint64_t announced = 0;
  if (!snappyLength(src, n, &ip, &announced) || announced != dstLen) {
    return false;
  }
  size_t op = 0;
  while (ip < n) {
    const unsigned tag = src[ip++];
    size_t len, offset;
    switch (tag & 3) {
      case 0: {
        len = (tag >> 2) + 1;
        if (len > 60) {
          const size_t extra = len - 60;   // 1..4 bytes of (length - 1), little endian
          if (ip + extra > n) {
            return false;
          }
          len = 0;
          for (size_t b = 0; b < extra; ++b) {
            len |= static_cast<size_t>(src[ip + b]) << (8 * b);
          }
          len += 1;
          ip += extra;
        }
        if (len > n - ip || len > dstLen - op) {
          return false;
        }
        std::memcpy(dst + op, src + ip, len);
        ip += len;
        op += len;
        continue;
      }
      case 1:
        if (ip + 1 > n) {
          return false;
        }
        len = ((tag >> 2) & 7) + 4;
        offset = (static_cast<size_t>(tag >> 5) << 8) | src[ip];
        ip += 1;
        break;
      case 2:
        if (ip + 2 > n) {
          return false;
        }
        len = (tag >> 2) + 1;
        offset = static_cast<size_t>(src[ip]) | (static_cast<size_t>(src[ip + 1]) << 8);
        ip += 2;
        break;
      default:
        if (ip + 4 > n) {
          return false;
        }
        len = (tag >> 2) + 1;
        offset = load32(src + ip);
        ip += 4;
        break;
    }
    if (offset == 0 || offset > op || len > dstLen - op) {
      return false;
    }
    for (size_t i = 0; i < len; ++i) {
      dst[op + i] = dst[op + i - offset];
    }
    op += len;
  }
  return op == dstLen;
}

void snappyLiteral(std::vector<unsigned char>& out, const unsigned char* lit, size_t len) {
  while (len) {
    const size_t piece = len;   // one literal element holds up to 2^32 bytes
    const size_t m = piece - 1;
    if (m < 60) {
      out.push_back(static_cast<unsigned char>(m << 2));
    } else {
      int bytes = m < (1u << 8) ? 1 : m < (1u << 16) ? 2 : m < (1u << 24) ? 3 : 4;
      out.push_back(static_cast<unsigned char>((59 + bytes) << 2));
      for (int b = 0; b < bytes; ++b) {
        out.push_back(static_cast<unsigned char>((m >> (8 * b)) & 0xff));
      }
    }
    out.insert(out.end(), lit, lit + piece);
    len -= piece;
  }
}

void snappyEncode(const unsigned char* src, size_t n, std::vector<unsigned char>& out) {
  out.clear();
  out.reserve(n / 2 + 16);
  for (uint64_t v = n;;) {
    if (v < 0x80) {
      out.push_back(static_cast<unsigned char>(v));
      break;
    }
    out.push_back(static_cast<unsigned char>((v & 0x7f) | 0x80));
    v >>= 7;
  }
  constexpr int kHashBits = 15;
  std::vector<uint32_t> table(size_t(1) << kHashBits, 0);
  size_t anchor = 0, i = 0;
  while (n >= 8 && i + 4 <= n) {
    const uint32_t h = (load32(src + i) * 0x1e35a7bdu) >> (32 - kHashBits);
    const size_t cand = table[h];
    table[h] = static_cast<uint32_t>(i + 1);
    if (cand && i + 1 - cand <= 65535 && load32(src + cand - 1) == load32(src + i)) {
      const size_t from = cand - 1;
      size_t len = 4;
      while (i + len < n && src[from + len] == src[i + len]) {
        ++len;
      }
      if (i > anchor) {
        snappyLiteral(out, src + anchor, i - anchor);
      }
      const size_t offset = i - from;
      for (size_t rest = len; rest;) {   // copies with a 2-byte offset hold 1..64 bytes
        const size_t piece = rest > 64 ? 64 : rest;
        out.push_back(static_cast<unsigned char>(((piece - 1) << 2) | 2));
        out.push_back(static_cast<unsigned char>(offset & 0xff));
        out.push_back(static_cast<unsigned char>(offset >> 8));
        rest -= piece;
      }
      i += len;
      anchor = i;
    } else {
      ++i;
    }
  }
  if (n > anchor) {
    snappyLiteral(out, src + anchor, n - anchor);
  }
}

// ---- zstd and zlib / gzip through the system libraries ------------------------------------------
struct ZstdApi {
  size_t (*decompress)(void*, size_t, const void*, size_t) = nullptr;
  size_t (*compress)(void*, size_t, const void*, size_t, int) = nullptr;
  size_t (*compressBound)(size_t) = nullptr;
  unsigned (*isError)(size_t) = nullptr;
  bool ok = false;
};

const ZstdApi& zstd() {
  static ZstdApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    void* lib = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
      lib = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL);
    }
    if (!lib) {
      return;
    }
    api.decompress = reinterpret_cast<decltype(api.decompress)>(dlsym(lib, "ZSTD_decompress"));
    api.compress = reinterpret_cast<decltype(api.compress)>(dlsym(lib, "ZSTD_compress"));
    api.compressBound = reinterpret_cast<decltype(api.compressBound)>(dlsym(lib, "ZSTD_compressBound"));
    api.isError = reinterpret_cast<decltype(api.isError)>(dlsym(lib, "ZSTD_isError"));
    api.ok = api.decompress && api.compress && api.compressBound && api.isError;
  });
  return api;
}

struct ZlibApi {
  int (*inflateInit2Fn)(z_streamp, int, const char*, int) = nullptr;
  int (*inflate)(z_streamp, int) = nullptr;
  int (*inflateEnd)(z_streamp) = nullptr;
  int (*deflateInit2Fn)(z_streamp, int, int, int, int, int, const char*, int) = nullptr;
  int (*deflate)(z_streamp, int) = nullptr;
  int (*deflateEnd)(z_streamp) = nullptr;
  uLong (*deflateBound)(z_streamp, uLong) = nullptr;
  bool ok = false;
};

const ZlibApi& zlibApi() {
  static ZlibApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    void* lib = dlopen("libz.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
      lib = dlopen("libz.so", RTLD_NOW | RTLD_LOCAL);
    }
    if (!lib) {
      return;
    }
    api.inflateInit2Fn = reinterpret_cast<decltype(api.inflateInit2Fn)>(dlsym(lib, "inflateInit2_"));
    api.inflate = reinterpret_cast<decltype(api.inflate)>(dlsym(lib, "inflate"));
    api.inflateEnd = reinterpret_cast<decltype(api.inflateEnd)>(dlsym(lib, "inflateEnd"));
    api.deflateInit2Fn = reinterpret_cast<decltype(api.deflateInit2Fn)>(dlsym(lib, "deflateInit2_"));
    api.deflate = reinterpret_cast<decltype(api.deflate)>(dlsym(lib, "deflate"));
    api.deflateEnd = reinterpret_cast<decltype(api.deflateEnd)>(dlsym(lib, "deflateEnd"));
    api.deflateBound = reinterpret_cast<decltype(api.deflateBound)>(dlsym(lib, "deflateBound"));
    api.ok = api.inflateInit2Fn && api.inflate && api.inflateEnd && api.deflateInit2Fn && api.deflate && api.deflateEnd &&
        api.deflateBound;
  });
  return api;
}

}  // namespace

const char* codecName(int32_t kind) {
  switch (kind) {
    case VX355_COMPRESSION_ZLIB:
      return "ZLIB";
    case VX355_COMPRESSION_SNAPPY:
      return "SNAPPY";
    case VX355_COMPRESSION_ZSTD:
      return "ZSTD";
    case VX355_COMPRESSION_LZ4:
      return "LZ4";
    case VX355_COMPRESSION_GZIP:
      return "GZIP";
    default:
      return nullptr;   // NONE, LZO, LZ4_HADOOP: "Not support ... in folly" (Compression.cpp:43)
  }
}

void codecUncompress(int32_t kind, const unsigned char* src, size_t n, unsigned char* dst, size_t dstLen) {
  bool good = false;
  switch (kind) {
    case VX355_COMPRESSION_LZ4:
      good = lz4Decode(src, n, dst, dstLen);
      break;
    case VX355_COMPRESSION_SNAPPY:
      good = snappyDecode(src, n, dst, dstLen);
      break;
    case VX355_COMPRESSION_ZSTD: {
      const ZstdApi& z = zstd();
      if (!z.ok) {
        VX_THROW(VX355_EUNSUPPORTED, "ZSTD pages need libzstd.so.1 on this host");
      }
      const size_t got = z.decompress(dst, dstLen, src, n);
      good = !z.isError(got) && got == dstLen;
      break;
    }
    case VX355_COMPRESSION_ZLIB:
    case VX355_COMPRESSION_GZIP: {
      const ZlibApi& z = zlibApi();
      if (!z.ok) {
        VX_THROW(VX355_EUNSUPPORTED, "ZLIB / GZIP pages need libz.so.1 on this host");
      }
      z_stream s;
      std::memset(&s, 0, sizeof(s));
      // 15 + 16: gzip wrapper, 15: zlib wrapper - what the configured kind says, as folly's codec does
      if (z.inflateInit2Fn(&s, kind == VX355_COMPRESSION_GZIP ? 15 + 16 : 15, ZLIB_VERSION, static_cast<int>(sizeof(z_stream))) != Z_OK) {
        VX_THROW(VX355_EINTERNAL, "inflateInit2 failed");
      }
      s.next_in = const_cast<Bytef*>(src);
      s.avail_in = static_cast<uInt>(n);
      s.next_out = dst;
      s.avail_out = static_cast<uInt>(dstLen);
      const int rc = z.inflate(&s, Z_FINISH);
      good = rc == Z_STREAM_END && s.avail_out == 0 && s.avail_in == 0;
      z.inflateEnd(&s);
      break;
    }
    default:
      VX_THROW(VX355_EUNSUPPORTED, "compression kind " + std::to_string(kind) + " (ZLIB, SNAPPY, ZSTD, LZ4, GZIP are the folly codecs Velox maps)");
  }
  if (!good) {
    // folly's codecs throw on a stream that is malformed or does not yield uncompressedSize bytes
    VX_THROW(VX355_EUSER, std::string("corrupt ") + codecName(kind) + " page body (or not the announced uncompressed size)");
  }
}

void codecCompress(int32_t kind, const unsigned char* src, size_t n, std::vector<unsigned char>& out) {
  switch (kind) {
    case VX355_COMPRESSION_LZ4:
      lz4Encode(src, n, out);
      return;
    case VX355_COMPRESSION_SNAPPY:
      snappyEncode(src, n, out);
      return;
    case VX355_COMPRESSION_ZSTD: {
      const ZstdApi& z = zstd();
      if (!z.ok) {
        VX_THROW(VX355_EUNSUPPORTED, "ZSTD pages need libzstd.so.1 on this host");
      }
      out.resize(z.compressBound(n));
      const size_t got = z.compress(out.data(), out.size(), src, n, 1);
      if (z.isError(got)) {
        VX_THROW(VX355_EINTERNAL, "ZSTD_compress failed");
      }
      out.resize(got);
      return;
    }
    case VX355_COMPRESSION_ZLIB:
    case VX355_COMPRESSION_GZIP: {
      const ZlibApi& z = zlibApi();
      if (!z.ok) {
        VX_THROW(VX355_EUNSUPPORTED, "ZLIB / GZIP pages need libz.so.1 on this host");
      }
      z_stream s;
      std::memset(&s, 0, sizeof(s));
      if (z.deflateInit2Fn(&s, Z_DEFAULT_COMPRESSION, Z_DEFLATED, kind == VX355_COMPRESSION_GZIP ? 15 + 16 : 15, 8,
                         Z_DEFAULT_STRATEGY, ZLIB_VERSION, static_cast<int>(sizeof(z_stream))) != Z_OK) {
        VX_THROW(VX355_EINTERNAL, "deflateInit2 failed");
      }
      out.resize(z.deflateBound(&s, static_cast<uLong>(n)) + 32);
      s.next_in = const_cast<Bytef*>(src);
      s.avail_in = static_cast<uInt>(n);
      s.next_out = out.data();
      s.avail_out = static_cast<uInt>(out.size());
      const int rc = z.deflate(&s, Z_FINISH);
      const size_t got = out.size() - s.avail_out;
      z.deflateEnd(&s);
      if (rc != Z_STREAM_END) {
        VX_THROW(VX355_EINTERNAL, "deflate failed");
      }
      out.resize(got);
      return;
    }
    default:
      VX_THROW(VX355_EUNSUPPORTED, "compression kind " + std::to_string(kind) + " (ZLIB, SNAPPY, ZSTD, LZ4, GZIP are the folly codecs Velox maps)");
  }
}

}  // namespace vx
