"""Pins the oracle's hash primitives against known-answer vectors and
transcribes the formula-style expectations of
/root/reference/velox/exec/tests/VectorHasherTest.cpp."""
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from velox_amd import abi

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
M64 = (1 << 64) - 1


def test_folly_known_answer_vectors(oracle):
    kat = json.load(open(os.path.join(GOLDEN, "folly_hash_kat.json")))
    L = oracle.lib()
    for x, want in kat["twang_mix64"]:
        assert L.orc_twang_mix64(int(x, 16)) == int(want)
    for x, want in kat["twang_32from64"]:
        assert L.orc_twang_32from64(int(x, 16)) == int(want)
    for x, want in kat["jenkins_rev_mix32"]:
        assert L.orc_jenkins_rev_mix32(int(x)) == int(want)


def _hash_mix_py(upper, lower):
    # bits::hashMix, common/base/BitUtil.h:775-784, in exact integer arithmetic.
    k = 0x9DDFEA08EB382D69
    a = ((lower ^ upper) * k) & M64
    a ^= a >> 47
    b = ((upper ^ a) * k) & M64
    b ^= b >> 47
    return (b * k) & M64


def test_hash_mix_matches_independent_python(oracle):
    rng = np.random.default_rng(1)
    L = oracle.lib()
    for _ in range(200):
        u, l = (int(x) for x in rng.integers(0, 1 << 63, 2, dtype=np.uint64) * 2 + 1)
        assert L.orc_hash_mix(u & M64, l & M64) == _hash_mix_py(u & M64, l & M64)


def _crc32c_bytes(crc, data):
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 & -(crc & 1))
    return crc & 0xFFFFFFFF


def test_crc32c_step(oracle):
    L = oracle.lib()
    # Standard CRC-32C check value: crc("123456789") == 0xE3069283 with
    # init/xorout 0xFFFFFFFF; the first 8 bytes go through one crc32U64 step.
    c = L.orc_crc32c_u64(0xFFFFFFFF, struct.unpack("<Q", b"12345678")[0])
    assert (_crc32c_bytes(c, b"9") ^ 0xFFFFFFFF) == 0xE3069283
    rng = np.random.default_rng(2)
    for _ in range(50):
        seed = int(rng.integers(0, 1 << 32))
        w = int(rng.integers(0, 1 << 63, dtype=np.uint64)) * 2 + int(rng.integers(0, 2))
        assert L.orc_crc32c_u64(seed, w) == _crc32c_bytes(seed, struct.pack("<Q", w))


@pytest.mark.skipif(not os.path.exists("/usr/bin/gcc"), reason="needs gcc")
def test_crc32c_step_matches_sse42_instruction(oracle):
    """The reference uses _mm_crc32_u64 on x86 (SimdUtil-inl.h:1401-1405); check the
    portable restatement against the real instruction when the host has it."""
    if "sse4_2" not in open("/proc/cpuinfo").read():
        pytest.skip("host CPU has no SSE4.2")
    src = r"""
    #include <nmmintrin.h>
    #include <stdio.h>
    #include <stdint.h>
    int main(){ uint64_t s=88172645463325252ULL; for(int i=0;i<64;i++){ s^=s<<13; s^=s>>7; s^=s<<17;
      uint32_t c=(uint32_t)(s>>11); uint64_t w=s*0x9E3779B97F4A7C15ULL;
      printf("%u %llu %u\n", c, (unsigned long long)w, (uint32_t)_mm_crc32_u64(c,w)); } return 0; }
    """
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-O1", "-msse4.2", "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split("\n")
    L = oracle.lib()
    for line in out:
        if line.strip():
            c, w, want = (int(x) for x in line.split())
            assert L.orc_crc32c_u64(c, w) == want


def test_xxh32_matches_xxhash_package(oracle):
    xxhash = pytest.importorskip("xxhash")
    L = oracle.lib()
    rng = np.random.default_rng(3)
    for v in [0, 1, 0xFFFFFFFF] + [int(x) for x in rng.integers(0, 1 << 32, 100)]:
        assert L.orc_xxh32_u32(v, 0) == xxhash.xxh32(struct.pack("<I", v), seed=0).intdigest()


def _hash_bytes_py(seed, data):
    # bits::hashBytes, common/base/BitUtil.cpp:177-225, bytewise independent restatement.
    kmul = 0x9DDFEA08EB382D69
    size = len(data)

    def crc64(c, word_bytes):
        return _crc32c_bytes(c & 0xFFFFFFFF, word_bytes.ljust(8, b"\0"))

    if size < 8:
        crc = crc64(seed, data)
        word = int.from_bytes(data.ljust(8, b"\0"), "little")
        crc2 = _crc32c_bytes(seed & 0xFFFFFFFF, struct.pack("<Q", word >> 32))
        return crc | (crc2 << 32)
    a0, a1, a2 = seed, (seed << 32) & M64, seed >> 16
    p = 0
    togo = size
    while togo >= 24:
        a0 = crc64(a0, data[p:p + 8])
        a1 = crc64(a1, data[p + 8:p + 16])
        a2 = crc64(a2, data[p + 16:p + 24])
        p += 24
        togo -= 24
    if togo > 16:
        a0 = crc64(a0, data[p:p + 8])
        a1 = crc64(a1, data[p + 8:p + 16])
        a2 = crc64(a2, data[p + 16:p + togo])
    elif togo > 8:
        a0 = crc64(a0, data[p:p + 8])
        a1 = crc64(a1, data[p + 8:p + togo])
    elif togo > 0:
        a0 = crc64(a0, data[p:p + togo])
    return a0 ^ ((a1 * kmul) & M64) ^ ((a2 * kmul) & M64)


def test_hash_bytes_all_lengths(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(4)
    for n in list(range(0, 60)) + [100, 255]:
        data = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert L.orc_hash_bytes(1, data, n) == _hash_bytes_py(1, data), n


def _folly_int64(L, v):
    return L.orc_twang_mix64(v & M64)


# --- VectorHasherTest.cpp:166-229 "flat" ------------------------------------
def test_flat_bigint_with_nulls_and_odd_rows(oracle):
    L = oracle.lib()
    values = np.arange(100, dtype=np.int64)
    valid = np.array([i % 5 != 0 for i in range(100)])
    batch = abi.HostBatch([abi.HostColumn(abi.BIGINT, values, valid)])
    odd = np.array([i % 2 == 1 for i in range(100)])
    hashes = oracle.hash_columns(batch, [0], rows=odd)
    for i in range(100):
        if i % 2 == 0:
            assert hashes[i] == 0
        elif i % 5 == 0:
            assert hashes[i] == 1  # kNullHash
        else:
            assert hashes[i] == _folly_int64(L, i)
    hashes = oracle.hash_columns(batch, [0])
    for i in range(100):
        assert hashes[i] == (1 if i % 5 == 0 else _folly_int64(L, i))
    # hashPrecomputed(mix=true) of 7 then 55: bits::hashMix(h(7), h(55)) (:219-228)
    c7 = abi.HostBatch([abi.HostColumn(abi.BIGINT, [7], encoding=abi.CONSTANT),
                        abi.HostColumn(abi.BIGINT, [55], encoding=abi.CONSTANT)], num_rows=10)
    mixed = oracle.hash_columns(c7, [0, 1])
    assert (mixed == L.orc_hash_mix(_folly_int64(L, 7), _folly_int64(L, 55))).all()


# --- VectorHasherTest.cpp:231-260 "nans" ------------------------------------
def test_nans_and_zeros(oracle):
    L = oracle.lib()
    snan = struct.unpack("<d", struct.pack("<Q", 0x7FF4000000000000))[0]
    vals = np.array([1.0, -1.0, float("nan"), snan, 0.0, -0.0])
    raw = vals.view(np.uint64).copy()
    raw[3] = 0x7FF4000000000000  # keep the signalling payload
    batch = abi.HostBatch([abi.HostColumn(abi.DOUBLE, raw.view(np.float64))])
    h = oracle.hash_columns(batch, [0])
    qnan_bits = struct.unpack("<Q", struct.pack("<d", float("nan")))[0]
    assert h[0] == L.orc_twang_mix64(struct.unpack("<Q", struct.pack("<d", 1.0))[0])
    assert h[1] == L.orc_twang_mix64(struct.unpack("<Q", struct.pack("<d", -1.0))[0])
    assert h[2] == h[3] == L.orc_twang_mix64(qnan_bits)
    assert h[4] == h[5] == 0


def test_small_ints_bool_strings_timestamp(oracle):
    L = oracle.lib()
    vals32 = np.array([0, 1, -1, 42, -2147483648, 2147483647], dtype=np.int32)
    for kind, dt in ((abi.INTEGER, np.int32), (abi.SMALLINT, np.int16), (abi.TINYINT, np.int8)):
        v = vals32.astype(dt)
        h = oracle.hash_columns(abi.HostBatch([abi.HostColumn(kind, v)]), [0])
        for x, got in zip(v, h):
            # folly integral_hasher: sign-extend to int32, jenkins_rev_mix32
            assert got == L.orc_jenkins_rev_mix32(int(x) & 0xFFFFFFFF)
    hb = oracle.hash_columns(abi.HostBatch([abi.HostColumn(abi.BOOLEAN, [True, False])]), [0])
    assert hb[0] == M64 and hb[1] == 0
    strs = [b"", b"a", b"abcdefg", b"abcdefgh", b"twelve_bytes", b"thirteen_byte", b"x" * 40]
    hs = oracle.hash_columns(abi.HostBatch([abi.HostColumn(abi.VARCHAR, strs)]), [0])
    for s, got in zip(strs, hs):
        assert got == _hash_bytes_py(1, s)
    ts = np.array([[5, 123456789], [-3, 0]], dtype=np.int64)
    ht = oracle.hash_columns(abi.HostBatch([abi.HostColumn(abi.TIMESTAMP, ts)]), [0])
    assert ht[0] == _hash_mix_py(5, 123456789)
    assert ht[1] == _hash_mix_py((-3) & M64, 0)


def test_dictionary_and_constant_encodings(oracle):
    L = oracle.lib()
    base = np.array([10, 20, 30], dtype=np.int64)
    idx = np.array([2, 0, 1, 1, 2, 0, 0], dtype=np.int32)
    valid = np.array([1, 1, 0, 1, 1, 1, 1], dtype=bool)
    col = abi.HostColumn(abi.BIGINT, base, valid, encoding=abi.DICTIONARY, indices=idx)
    h = oracle.hash_columns(abi.HostBatch([col]), [0])
    for i in range(7):
        assert h[i] == (1 if not valid[i] else _folly_int64(L, int(base[idx[i]])))
    cnull = abi.HostColumn(abi.BIGINT, [0], valid=[False], encoding=abi.CONSTANT)
    h = oracle.hash_columns(abi.HostBatch([cnull], num_rows=5), [0])
    assert (h == 1).all()


# --- VectorHasherTest.cpp:82-145 testComputeValueIds --------------------------
@pytest.mark.parametrize("kind,dt", [(abi.BIGINT, np.int64), (abi.INTEGER, np.int32),
                                     (abi.SMALLINT, np.int16), (abi.TINYINT, np.int8)])
@pytest.mark.parametrize("with_nulls", [False, True])
@pytest.mark.parametrize("multiplier", [1, 27])
def test_compute_value_ids_range(oracle, kind, dt, with_nulls, multiplier):
    size = 111
    info = np.iinfo(dt)
    for lo in (0, int(info.min), int(info.max) - 16):
        vals = np.array([lo + i % 17 for i in range(size)], dtype=dt)
        valid = np.array([i % 7 != 0 for i in range(size)]) if with_nulls else None
        col = abi.HostColumn(kind, vals, valid)
        h = oracle.Hasher(kind)
        ok, _ = h.compute_value_ids(col, rows=np.ones(size, bool))
        assert not ok
        assert h.cardinality(0) == (18, 18)
        assert h.enable_value_range(multiplier, 0) == 18 * multiplier
        result = np.zeros(size, dtype=np.uint64)
        ok, result = h.compute_value_ids(col, rows=np.ones(size, bool), result=result)
        assert ok
        for i in range(size):
            if valid is not None and not valid[i]:
                assert result[i] == 0
            else:
                assert result[i] == (i % 17 + 1) * multiplier
        odd = np.array([i % 2 == 1 for i in range(size)])
        result = np.zeros(size, dtype=np.uint64)
        ok, result = h.compute_value_ids(col, rows=odd, result=result)
        assert ok
        for i in range(size):
            if i % 2 == 0 or (valid is not None and not valid[i]):
                assert result[i] == 0
            else:
                assert result[i] == (i % 17 + 1) * multiplier
        if lo == 0:
            out_of_range = abi.HostColumn(kind, np.array([i % 19 for i in range(size)], dtype=dt), valid)
            ok, _ = h.compute_value_ids(out_of_range, rows=np.ones(size, bool))
            assert not ok
            r, d = h.cardinality(0)
            assert r > 18 and d > 18


def test_string_value_ids_and_string_as_number(oracle):
    # VectorHasher.h:383-387 stringAsNumber: bytes as little endian + 1 << (8*size)
    strs = [b"A", b"N", b"R", b"N", b"A"]
    col = abi.HostColumn(abi.VARCHAR, strs)
    h = oracle.Hasher(abi.VARCHAR)
    ok, _ = h.compute_value_ids(col, rows=np.ones(5, bool))
    assert not ok
    st = h.state()
    assert st.min == 256 + ord("A") and st.max == 256 + ord("R")
    assert h.cardinality(0) == (ord("R") - ord("A") + 2, 4)
    # group-by reserve of 50 %: 2 + 17 * 0.5 = 10 on each side (VectorHasher.cpp:786-835)
    assert h.cardinality(50)[0] == (ord("R") - ord("A")) + 2 * 10 + 2
    h.enable_value_range(1, 50)
    ok, ids = h.compute_value_ids(col, rows=np.ones(5, bool))
    assert ok
    base = 256 + ord("A") - 10
    assert list(ids) == [256 + ord(s) - base + 1 for s in (b"A", b"N", b"R", b"N", b"A")]
    # 8-byte strings cannot be a range
    h2 = oracle.Hasher(abi.VARCHAR)
    h2.compute_value_ids(abi.HostColumn(abi.VARCHAR, [b"12345678"]), rows=np.ones(1, bool))
    assert h2.cardinality(0)[0] == M64  # kRangeTooLarge


def test_distinct_value_ids_and_lookup(oracle):
    vals = np.array([1000000, 5, 1000000, -7, 5, 123456789012], dtype=np.int64)
    col = abi.HostColumn(abi.BIGINT, vals)
    h = oracle.Hasher(abi.BIGINT)
    h.compute_value_ids(col, rows=np.ones(6, bool))
    assert h.state().num_distinct == 4
    assert h.enable_value_ids(1, 50) == 4 * 1.5 + 1
    ok, ids = h.compute_value_ids(col, rows=np.ones(6, bool))
    assert ok and list(ids) == [1, 2, 1, 3, 2, 4]  # ids in first-seen order
    probe = abi.HostColumn(abi.BIGINT, np.array([5, 6, -7], dtype=np.int64))
    rows, ids = h.lookup_value_ids(probe, np.ones(3, bool))
    assert list(rows) == [True, False, True]
    assert ids[0] == 2 and ids[2] == 3


def test_boolean_value_ids(oracle):
    col = abi.HostColumn(abi.BOOLEAN, [True, False, True], valid=[True, True, False])
    h = oracle.Hasher(abi.BOOLEAN)
    assert h.cardinality(0) == (3, 3)
    assert h.enable_value_range(1, 0) == 3
    ok, ids = h.compute_value_ids(col, rows=np.ones(3, bool))
    assert ok and list(ids) == [2, 1, 0]


# --- VectorHasherTest.cpp:670-770 merge -------------------------------------
def test_merge_ranges_and_distincts(oracle):
    a, b = oracle.Hasher(abi.BIGINT), oracle.Hasher(abi.BIGINT)
    a.compute_value_ids(abi.HostColumn(abi.BIGINT, np.arange(1, 101, dtype=np.int64)), rows=np.ones(100, bool))
    b.compute_value_ids(abi.HostColumn(abi.BIGINT, np.arange(50, 201, dtype=np.int64)), rows=np.ones(151, bool))
    a.merge(b)
    st = a.state()
    assert (st.min, st.max, st.num_distinct) == (1, 200, 200)
    assert a.cardinality(0) == (201, 201)
