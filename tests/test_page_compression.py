"""Compressed PrestoPages (PrestoOptions::compressionKind; serializers/PrestoSerializer.cpp:144-145,185-199,
PrestoSerializerSerializationUtils.h:279-334). vx355_presto_compress_page / _uncompress_page are pure
host functions of the C ABI, so this file runs without a GPU: pages come from the oracle's writer, and
every codec is checked against an INDEPENDENT implementation of the published format (pyarrow's lz4_raw /
snappy / zstd, Python's zlib for RFC 1950 / 1952) in both directions."""
import struct
import zlib

import numpy as np
import pyarrow as pa
import pytest

import oracle_lib as oracle
from presto_page_reader import random_page_batch
from velox_amd import abi, ops

KINDS = {"LZ4": abi.COMPRESSION_LZ4, "SNAPPY": abi.COMPRESSION_SNAPPY, "ZSTD": abi.COMPRESSION_ZSTD,
         "ZLIB": abi.COMPRESSION_ZLIB, "GZIP": abi.COMPRESSION_GZIP}


def independent_compress(name, body):
    if name == "ZLIB":
        return zlib.compress(body)
    if name == "GZIP":
        c = zlib.compressobj(wbits=31)
        return c.compress(body) + c.flush()
    return pa.compress(body, codec={"LZ4": "lz4_raw", "SNAPPY": "snappy", "ZSTD": "zstd"}[name], asbytes=True)


def independent_uncompress(name, data, size):
    if name == "ZLIB":
        return zlib.decompress(data)
    if name == "GZIP":
        return zlib.decompress(data, wbits=31)
    return pa.decompress(data, decompressed_size=size, codec={"LZ4": "lz4_raw", "SNAPPY": "snappy", "ZSTD": "zstd"}[name],
                         asbytes=True)


def header(page):
    n, codec, uncompressed, stored, checksum = struct.unpack_from("<ibiiq", page, 0)
    return n, codec & 0xff, uncompressed, stored, checksum


def checksum_of(body, codec, n, uncompressed):
    """computeChecksum (PrestoSerializer.cpp:39-79) with Python's CRC-32."""
    crc = zlib.crc32(body)
    crc = zlib.crc32(bytes([codec]), crc)
    crc = zlib.crc32(struct.pack("<i", n), crc)
    return zlib.crc32(struct.pack("<i", uncompressed), crc)


def assemble(page, name, body):
    """The page a compressing writer sends: header with the compressed bit around an independently compressed body."""
    n, codec, uncompressed, _, _ = header(page)
    marked = codec | 1
    check = checksum_of(body, marked, n, uncompressed) if marked & 4 else 0
    return struct.pack("<ibiiq", n, marked, uncompressed, len(body), check) + body


def oracle_pages(seed, rows, flags):
    rng = np.random.default_rng(seed)
    batch, _ = random_page_batch(rng, rows)
    return [p for p in oracle.presto_serialize(batch, [0, rows // 3, rows], flags=flags) if p]


@pytest.mark.parametrize("name", sorted(KINDS))
@pytest.mark.parametrize("flags", [0, abi.PAGE_CHECKSUM])
def test_compressed_pages_decode_with_an_independent_codec_and_round_trip(name, flags):
    kind = KINDS[name]
    for page in oracle_pages(11, 4000, flags):
        n, codec, uncompressed, stored, check = header(page)
        assert uncompressed == stored == len(page) - 21
        if flags:
            assert codec & 4 and check == checksum_of(page[21:], codec, n, uncompressed)   # the CRC convention itself
        packed = ops.presto_compress_page(page, kind)
        pn, pcodec, puncompressed, pstored, pcheck = header(packed)
        assert (pn, puncompressed) == (n, uncompressed) and pcodec == (codec | 1)
        assert pstored == len(packed) - 21 and pstored <= 0.8 * uncompressed
        assert independent_uncompress(name, packed[21:], uncompressed) == page[21:]
        assert pcheck == (checksum_of(packed[21:], pcodec, n, uncompressed) if flags else 0)
        assert ops.presto_uncompress_page(packed, kind) == page
        # the other direction: a body compressed by the independent codec
        foreign = assemble(page, name, independent_compress(name, page[21:]))
        assert ops.presto_uncompress_page(foreign, kind) == page
        # an uncompressed page passes through both functions unchanged
        assert ops.presto_uncompress_page(page, kind) == page
        assert ops.presto_compress_page(page, abi.COMPRESSION_NONE) == page


@pytest.mark.parametrize("name", sorted(KINDS))
def test_pages_that_do_not_compress_travel_uncompressed(name):
    """flushCompressed keeps the uncompressed page when compressedSize > uncompressedSize * minCompressionRatio."""
    rng = np.random.default_rng(5)
    noise = rng.integers(-2**62, 2**62, 3000).astype(np.int64)
    batch = abi.HostBatch([abi.HostColumn(abi.BIGINT, noise)])
    page, = oracle.presto_serialize(batch, [0, 3000], flags=abi.PAGE_CHECKSUM)
    assert ops.presto_compress_page(page, KINDS[name]) == page
    # ... and a ratio nothing reaches keeps even a compressible page as it is
    easy = abi.HostBatch([abi.HostColumn(abi.BIGINT, np.zeros(3000, dtype=np.int64))])
    page, = oracle.presto_serialize(easy, [0, 3000])
    assert ops.presto_compress_page(page, KINDS[name], min_ratio=1e-6) == page
    assert len(ops.presto_compress_page(page, KINDS[name])) < len(page) // 4


def test_corrupt_compressed_pages_are_user_errors():
    page = oracle_pages(12, 3000, abi.PAGE_CHECKSUM)[-1]
    plain = oracle_pages(12, 3000, 0)[-1]
    for name, kind in KINDS.items():
        packed = bytearray(ops.presto_compress_page(page, kind))
        packed[21 + len(packed) // 2] ^= 0x40
        with pytest.raises(ops.Vx355Error) as e:      # the checksum covers the compressed bytes
            ops.presto_uncompress_page(bytes(packed), kind)
        assert e.value.status == abi.EUSER and "corrupted" in str(e.value)
        # without a checksum the codec itself (or the size it must produce) catches it
        packed = ops.presto_compress_page(plain, kind)
        for broken in (packed[:21] + packed[21:len(packed) - 7], packed[:21] + packed[28:]):
            broken = struct.pack("<ibiiq", *header(broken)[:3], len(broken) - 21, 0) + broken[21:]
            with pytest.raises(ops.Vx355Error) as e:
                ops.presto_uncompress_page(broken, kind)
            assert e.value.status == abi.EUSER, name
        # the announced uncompressed size is part of the contract
        n, codec, uncompressed, stored, _ = header(packed)
        lying = struct.pack("<ibiiq", n, codec, uncompressed + 8, stored, 0) + packed[21:]
        with pytest.raises(ops.Vx355Error) as e:
            ops.presto_uncompress_page(lying, kind)
        assert e.value.status == abi.EUSER
    packed = ops.presto_compress_page(plain, abi.COMPRESSION_LZ4)
    with pytest.raises(ops.Vx355Error) as e:          # the reader's configuration names the codec
        ops.presto_uncompress_page(packed, abi.COMPRESSION_NONE)
    assert e.value.status == abi.EINVAL
    with pytest.raises(ops.Vx355Error) as e:          # no folly codec for LZO (Compression.cpp:43)
        ops.presto_compress_page(plain, abi.COMPRESSION_LZO)
    assert e.value.status == abi.EUNSUPPORTED
    with pytest.raises(ops.Vx355Error) as e:
        ops.presto_compress_page(packed, abi.COMPRESSION_LZ4)   # already compressed
    assert e.value.status == abi.EINVAL


def test_own_lz4_and_snappy_coders_on_awkward_inputs():
    """The two codecs the library implements itself: sizes around the formats' end conditions (LZ4: last 5
    bytes literal, last match 12 bytes before the end), long runs (length bytes of 255), overlapping
    matches, and garbage that must be rejected without reading outside the buffers."""
    rng = np.random.default_rng(99)
    bodies = [bytes(rng.integers(0, 4, n).astype(np.uint8)) for n in list(range(0, 40)) + [255, 256, 270, 4096, 70000]]
    bodies += [b"a" * n for n in (13, 18, 19, 20, 274, 275, 600, 66000)] + [b"abcd" * 5000 + bytes(range(256)) * 3]
    for body in bodies:
        page = struct.pack("<ibiiq", 1, 0, len(body), len(body), 0) + body
        for name in ("LZ4", "SNAPPY"):
            packed = ops.presto_compress_page(page, KINDS[name], min_ratio=10.0)
            if packed != page:
                assert independent_uncompress(name, packed[21:], len(body)) == body, (name, len(body))
                assert ops.presto_uncompress_page(packed, KINDS[name]) == page
            if len(body) > 0:
                theirs = independent_compress(name, body)
                if len(theirs) < len(body):
                    assert ops.presto_uncompress_page(assemble(page, name, theirs), KINDS[name]) == page
    for trial in range(300):
        junk = bytes(rng.integers(0, 256, int(rng.integers(1, 200))).astype(np.uint8))
        size = int(rng.integers(len(junk) + 1, 4000))
        for name in ("LZ4", "SNAPPY"):
            page = struct.pack("<ibiiq", 1, 1, size, len(junk), 0) + junk
            try:
                out = ops.presto_uncompress_page(page, KINDS[name])
                assert len(out) == 21 + size     # (random bytes that happen to be a valid stream of that size)
            except ops.Vx355Error as e:
                assert e.status == abi.EUSER
