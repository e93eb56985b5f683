"""HashBuild / HashProbe on the MI355X vs the CPU oracle, through the C ABI.

Bit-exact bar: the set of (probe row, build row) pairs, payload values and
validity, ascending probe-row order with all matches of one probe row
contiguous (HashTable::listJoinResults contract). The ORDER of the matches
inside one probe row follows the duplicate chain, which the reference itself
does not keep stable (parallel build): it is compared as a set."""
import numpy as np
import pytest

from velox_amd import abi
from gpu_util import batch_of

pytestmark = pytest.mark.gpu

JOIN_TYPES = [abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_LEFT_SEMI_FILTER, abi.JOIN_ANTI]


def _drain(probe, max_rows, build_col_ids=None):
    pairs, payload = [], []
    last = -1
    while True:
        mapping, build_rows, cols, fin = probe.get_output(max_rows, build_col_ids)
        assert len(mapping) <= max_rows
        assert (np.diff(mapping) >= 0).all() and (len(mapping) == 0 or mapping[0] >= last)
        if len(mapping):
            last = int(mapping[-1])
        for i in range(len(mapping)):
            pairs.append((int(mapping[i]), int(build_rows[i])))
            row = []
            for vals, valid in cols:
                if not valid[i]:
                    row.append(None)
                elif isinstance(vals, list):
                    row.append(vals[i])
                else:
                    row.append(vals[i].item() if hasattr(vals[i], "item") else vals[i])
            payload.append(tuple(row))
        if fin:
            break
    return pairs, payload


def _canon(pairs, payload):
    return sorted(zip(pairs, payload), key=lambda t: (t[0][0], t[0][1]))


def _contiguous(pairs):
    seen, prev = set(), None
    for r, _ in pairs:
        if r != prev:
            assert r not in seen
            seen.add(r)
            prev = r


def _build(impl, batches_per_driver, key_cols, key_types, dep_cols, dep_types, join_type):
    builds = []
    for batches in batches_per_driver:
        b = impl.JoinBuild(key_cols, key_types, dep_cols, dep_types, join_type)
        for hb in batches:
            b.add_input(hb)
        builds.append(b)
    return builds[0].finish(builds[1:]), builds


@pytest.mark.parametrize("join_type", JOIN_TYPES)
@pytest.mark.parametrize("force_hash", [False, True])
def test_join_duplicates_nulls_two_drivers(oracle, vx, join_type, force_hash, monkeypatch):
    if force_hash:
        monkeypatch.setenv("VX355_JOIN_ARRAY_MAX", "0")
    rng = np.random.default_rng(11)
    nb, npb = 6000, 20000
    bk = rng.integers(0, 1500, nb).astype(np.int64)
    bvalid = rng.random(nb) > 0.05
    bpay = rng.integers(0, 1 << 40, nb).astype(np.int64)
    bpay2 = rng.random(nb)
    bpay2_valid = rng.random(nb) > 0.3
    bstr = [bytes(rng.integers(65, 91, int(rng.integers(0, 13))).astype(np.uint8)) for _ in range(nb)]
    pk = rng.integers(-100, 3000, npb).astype(np.int64)
    pvalid = rng.random(npb) > 0.05

    def driver(lo, hi):
        out = []
        for s in range(lo, hi, 1000):
            e = min(hi, s + 1000)
            out.append(abi.HostBatch([abi.HostColumn(abi.BIGINT, bk[s:e], bvalid[s:e]),
                                      abi.HostColumn(abi.BIGINT, bpay[s:e]),
                                      abi.HostColumn(abi.DOUBLE, bpay2[s:e], bpay2_valid[s:e]),
                                      abi.HostColumn(abi.VARCHAR, bstr[s:e])]))
        return out
    deps, dep_types = [1, 2, 3], [abi.BIGINT, abi.DOUBLE, abi.VARCHAR]
    results = {}
    for impl in (oracle, vx):
        table, builds = _build(impl, [driver(0, 3500), driver(3500, nb)], [0], [abi.BIGINT], deps,
                               dep_types, join_type)
        st = table.stats()
        assert st.num_rows == bvalid.sum() and st.has_duplicates == 1
        assert st.num_distinct == len(np.unique(bk[bvalid]))
        probe = impl.JoinProbe(table, [0], join_type)
        probe.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, pk, pvalid)]))
        pairs, payload = _drain(probe, 1000 if impl is oracle else 777)
        _contiguous(pairs)
        results[impl.__name__] = _canon(pairs, payload)
        if impl is vx:
            assert st.hash_mode == (abi.MODE_NORMALIZED_KEY if force_hash else abi.MODE_ARRAY)
    assert results[oracle.__name__] == results[vx.__name__]


def test_join_unique_keys_probe_twice_and_no_payload(oracle, vx):
    rng = np.random.default_rng(21)
    nb = 50000
    bk = rng.permutation(200000)[:nb].astype(np.int64)
    pay = rng.integers(0, 1000, nb).astype(np.int32)
    res = {}
    for impl in (oracle, vx):
        table, builds = _build(impl, [[batch_of([bk, pay])]], [0], [abi.BIGINT], [1], [abi.INTEGER],
                               abi.JOIN_INNER)
        assert table.stats().has_duplicates == 0 and table.stats().num_distinct == nb
        probe = impl.JoinProbe(table, [0], abi.JOIN_INNER)
        out = []
        for seed in (1, 2):
            pk = np.random.default_rng(seed).integers(0, 220000, 70000).astype(np.int64)
            probe.add_input(batch_of([pk]))
            pairs, payload = _drain(probe, 65536)
            out.append(_canon(pairs, payload))
            # no payload columns requested
            probe.add_input(batch_of([pk]))
            pairs2, payload2 = _drain(probe, 100000, build_col_ids=[])
            assert sorted(pairs2) == sorted(pairs) and all(p == () for p in payload2)
        res[impl.__name__] = out
    assert res[oracle.__name__] == res[vx.__name__]


def test_join_two_keys_strings_and_small_ints(oracle, vx):
    rng = np.random.default_rng(31)
    nb, npb = 3000, 8000
    segs = [b"AUTOMOB", b"BUILDIN", b"FURNITU", b"MACHINE", b"HOUSEHO", b""]
    bk1 = [segs[i] for i in rng.integers(0, 6, nb)]
    bk2 = rng.integers(-20, 20, nb).astype(np.int16)
    pay = np.arange(nb, dtype=np.int64)
    pk1 = [segs[i] if i < 6 else b"TOOLONG8" for i in rng.integers(0, 7, npb)]
    pk2 = rng.integers(-25, 25, npb).astype(np.int16)
    pvalid = rng.random(npb) > 0.1
    res = {}
    for impl in (oracle, vx):
        table, _b = _build(impl, [[batch_of([bk1, bk2, pay])]], [0, 1], [abi.VARCHAR, abi.SMALLINT], [2],
                           [abi.BIGINT], abi.JOIN_LEFT)
        probe = impl.JoinProbe(table, [0, 1], abi.JOIN_LEFT)
        probe.add_input(batch_of([pk1, pk2], [pvalid, None]))
        pairs, payload = _drain(probe, 512)
        res[impl.__name__] = sorted(zip([p[0] for p in pairs], payload))
    assert res[oracle.__name__] == res[vx.__name__]


def test_join_empty_build_and_empty_probe(oracle, vx):
    for impl in (oracle, vx):
        b = impl.JoinBuild([0], [abi.BIGINT], [], [], abi.JOIN_LEFT)
        b.add_input(batch_of([np.zeros(0, dtype=np.int64)]))
        b.add_input(batch_of([np.array([5, 6], dtype=np.int64)], [np.array([False, False])]))
        table = b.finish()
        assert table.stats().num_rows == 0
        probe = impl.JoinProbe(table, [0], abi.JOIN_LEFT)
        probe.add_input(batch_of([np.array([1, 2, 3], dtype=np.int64)]))
        mapping, rows, cols, fin = probe.get_output(10, [])
        assert list(mapping) == [0, 1, 2] and list(rows) == [-1, -1, -1] and fin
        probe.add_input(batch_of([np.zeros(0, dtype=np.int64)]))
        mapping, rows, cols, fin = probe.get_output(10, [])
        assert len(mapping) == 0 and fin


def test_unsupported_join_kinds_are_refused(vx):
    with pytest.raises(vx.Vx355Error) as e:
        vx.JoinBuild([0], [abi.BIGINT], [], [], abi.JOIN_RIGHT_SEMI_PROJECT, null_aware=True)
    assert e.value.status == abi.EUNSUPPORTED
    with pytest.raises(vx.Vx355Error) as e:
        vx.JoinBuild([0], [abi.BIGINT], [], [], abi.JOIN_LEFT, null_aware=True)
    assert e.value.status == abi.EUNSUPPORTED
    with pytest.raises(vx.Vx355Error) as e:
        vx.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_COUNTING_ANTI)   # counting joins carry no payload
    assert e.value.status == abi.EINVAL


def test_q3_shape_device_resident_probe(oracle, vx):
    """TPC-H Q3 shape scaled down: sparse order keys, selective probe, outputs
    left in HBM."""
    rng = np.random.default_rng(41)
    n_orders, n_line = 60000, 400000
    okeys = (np.arange(n_orders, dtype=np.int64) // 8) * 32 + (np.arange(n_orders) % 8)
    build_sel = rng.random(n_orders) < 0.1
    bk = okeys[build_sel]
    odate = rng.integers(8000, 10000, len(bk)).astype(np.int32)
    lk = okeys[rng.integers(0, n_orders, n_line)]
    table_o, _bo = _build(oracle, [[batch_of([bk, odate])]], [0], [abi.BIGINT], [1], [abi.INTEGER], abi.JOIN_INNER)
    po = oracle.JoinProbe(table_o, [0], abi.JOIN_INNER)
    po.add_input(batch_of([lk]))
    e_pairs, e_payload = _drain(po, 1 << 20)
    table, _b = _build(vx, [[vx.to_device(batch_of([bk, odate]))]], [0], [abi.BIGINT], [1], [abi.INTEGER],
                       abi.JOIN_INNER)
    probe = vx.JoinProbe(table, [0], abi.JOIN_INNER)
    probe.add_input(vx.to_device(batch_of([lk])))
    cap = len(e_pairs) + 10
    mapping, rows = vx.DeviceArray(cap, np.int32), vx.DeviceArray(cap, np.int32)
    dates, dnulls = vx.DeviceArray(cap, np.int32), vx.DeviceArray((cap + 63) // 64, np.uint64)
    descs = (abi.OutColumn * 1)()
    descs[0].type_kind, descs[0].mem = abi.INTEGER, abi.MEM_DEVICE
    descs[0].values, descs[0].nulls = dates.ptr, dnulls.ptr
    n, fin = probe.get_output_device(cap, mapping.ptr, rows.ptr, descs, [0])
    assert fin and n == len(e_pairs)
    got = list(zip(mapping.to_host(n).tolist(), rows.to_host(n).tolist()))
    assert got == e_pairs
    assert dates.to_host(n).tolist() == [p[0] for p in e_payload]


@pytest.mark.parametrize("hit_rate,windowed,wide", [(1.0, False, "1"), (0.9, True, "1"), (0.02, False, "1"), (1.0, False, "-1"), (1.0, False, "0")])
def test_inline_dependents_of_wide_slots(oracle, vx, hit_rate, windowed, wide, monkeypatch):
    """Inner join, unique sparse build keys (normalized-key mode), outputs left in HBM: the probe finds
    up to two 8-byte dependents in the slot next to the key (WideSlot) and the emit pass writes them,
    the other build columns (a nullable BIGINT, an INTEGER, a third 8-byte column) are gathered by build
    row - high hit rates (every tile dense), a low one (tiles listed by the probe pass), output taken
    in windows, the adaptive rule (-1: this batch is 8 x the build side) and the switch off (0)."""
    monkeypatch.setenv("VX355_JOIN_WIDE", wide)
    rng = np.random.default_rng(77)
    nb, npr = 50_000, (1_200_000 if wide == "-1" else 400_000)   # adaptive: batches of >= 2^20 rows, >= 2 x the build
    bk = (rng.permutation(1 << 22)[:nb].astype(np.int64) * 104729 + 12345) * 1_000_003   # sparse: no array mode
    d0 = rng.integers(-1 << 60, 1 << 60, nb).astype(np.int64)
    d1 = rng.random(nb)
    d2 = rng.integers(0, 1 << 40, nb).astype(np.int64)
    d2_valid = rng.random(nb) > 0.2
    d3 = rng.integers(0, 1 << 30, nb).astype(np.int32)
    d4 = rng.integers(0, 1 << 50, nb).astype(np.int64)
    lk = np.where(rng.random(npr) < hit_rate, bk[rng.integers(0, nb, npr)], -7).astype(np.int64)
    dep_kinds = [abi.BIGINT, abi.DOUBLE, abi.BIGINT, abi.INTEGER, abi.BIGINT]
    build = batch_of([bk, d0, d1, d2, d3, d4], [None, None, None, d2_valid, None, None])
    table_o, _bo = _build(oracle, [[build]], [0], [abi.BIGINT], [1, 2, 3, 4, 5], dep_kinds, abi.JOIN_INNER)
    po = oracle.JoinProbe(table_o, [0], abi.JOIN_INNER)
    po.add_input(batch_of([lk]))
    e_map, e_rows, e_cols = [], [], None
    while True:
        m, r, cols, fin = po.get_output(1 << 20)
        e_map += list(m)
        e_rows += list(r)
        e_cols = cols if e_cols is None else [(np.concatenate([a[0], b[0]]), np.concatenate([a[1], b[1]])) for a, b in zip(e_cols, cols)]
        if fin:
            break
    table, _b = _build(vx, [[vx.to_device(build)]], [0], [abi.BIGINT], [1, 2, 3, 4, 5], dep_kinds, abi.JOIN_INNER)
    assert table.stats().hash_mode == abi.MODE_NORMALIZED_KEY
    vx.profile_reset()
    vx.profile_enable(True)
    probe = vx.JoinProbe(table, [0], abi.JOIN_INNER)
    probe.add_input(vx.to_device(batch_of([lk])))
    cap = 70_000 if windowed else len(e_map) + 64
    mapping, rows = vx.DeviceArray(cap, np.int32), vx.DeviceArray(cap, np.int32)
    np_types = [np.int64, np.float64, np.int64, np.int32, np.int64]
    vals = [vx.DeviceArray(cap, t) for t in np_types]
    nulls = [vx.DeviceArray((cap + 63) // 64, np.uint64) for _ in np_types]
    order = [4, 1, 0, 2, 3]   # not the table's order
    descs = (abi.OutColumn * 5)()
    for i, c in enumerate(order):
        descs[i].type_kind, descs[i].mem = dep_kinds[c], abi.MEM_DEVICE
        descs[i].values, descs[i].nulls = vals[c].ptr, nulls[c].ptr
    got_map, got_rows, got_cols = [], [], [[] for _ in np_types]
    got_valid = [[] for _ in np_types]
    while True:
        n, fin = probe.get_output_device(cap, mapping.ptr, rows.ptr, descs, order)
        got_map += mapping.to_host(n).tolist()
        got_rows += rows.to_host(n).tolist()
        for c in range(5):
            got_cols[c] += vals[c].to_host(n).tolist()
            words = nulls[c].to_host((n + 63) // 64)
            got_valid[c] += [bool((int(words[i >> 6]) >> (i & 63)) & 1) for i in range(n)]
            if n % 64:
                assert int(words[-1]) >> (n % 64) == 0   # bits past the page stay clear, as the gather leaves them
        if fin:
            break
    vx.profile_enable(False)
    assert got_map == [int(x) for x in e_map] and got_rows == [int(x) for x in e_rows]
    for c in range(5):
        ev, evalid = e_cols[c]
        evalid = np.asarray(evalid).astype(bool).tolist()
        assert got_valid[c] == evalid, c
        ev = np.asarray(ev)
        if ev.dtype.kind == "f":
            assert [np.float64(g).view(np.int64) for g, ok in zip(got_cols[c], evalid) if ok] == \
                [np.float64(e).view(np.int64) for e, ok in zip(ev.tolist(), evalid) if ok]
        else:
            assert [g for g, ok in zip(got_cols[c], evalid) if ok] == [e for e, ok in zip(ev.tolist(), evalid) if ok], c
    prof = vx.profile()
    assert ("k_widen_slots" in prof) == (wide != "0")


def test_repartitioned_join_gpu_backend_single_rank(oracle, vx):
    """The config-5 pipeline on one GPU (world = 1, RCCL): hash -> partition ->
    stable scatter -> all-to-all -> local build + probe, all in HBM, against the
    oracle's join of the same rows."""
    import os
    import torch
    import torch.distributed as dist
    from velox_amd import dist as vdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        rng = np.random.default_rng(51)
        nd, nf = 50000, 400000
        pk = rng.permutation(1 << 20)[:nd].astype(np.int64)
        a = rng.integers(0, 1 << 40, nd).astype(np.int64)
        fk = np.where(rng.random(nf) < 0.9, pk[rng.integers(0, nd, nf)], -1).astype(np.int64)
        t = [torch.from_numpy(x).to(dev) for x in (pk, a, fk)]
        backend = vdist.GpuJoinBackend(vx, torch)
        total, outputs, stats = vdist.repartitioned_join(backend, dist, torch, [t[0], t[1]], [t[2]])
        assert stats.num_rows == nd
        lookup = dict(zip(pk.tolist(), a.tolist()))
        want = sorted(lookup[k] for k in fk.tolist() if k in lookup)
        got = sorted(np.concatenate([p.cpu().numpy() for _, p in outputs]).tolist())
        assert total == len(want) and got == want
        # the pipelined form (probe side in chunks, asynchronous all-to-all) gives the same rows
        per_chunk, table = vdist.repartitioned_join_pipelined(backend, dist, torch, [t[0], t[1]], [t[2]], chunks=3)
        assert len(per_chunk) == 3 and table.stats().num_rows == nd
        got = sorted(np.concatenate([p.cpu().numpy() for _, outs in per_chunk for _, p in outs]).tolist())
        assert got == want
        for received, outs in per_chunk:      # mappings index the rows of their own chunk
            for m, p in outs:
                keys = received[0][m.long()].cpu().numpy().tolist()
                assert [lookup[k] for k in keys] == p.cpu().numpy().tolist()
        # the same exchange through the library's own RCCL communicator (vx355_comm_*,
        # vx355_exchange_*): what a C++ host calls; torch.distributed is not involved
        comm = vx.Comm(vx.Comm.unique_id(), 1, 0)
        total2, outputs2, _ = vdist.repartitioned_join(backend, dist, torch, [t[0], t[1]], [t[2]],
                                                      exchange_fn=vdist.LibExchange(torch, comm))
        got = sorted(np.concatenate([p.cpu().numpy() for _, p in outputs2]).tolist())
        assert total2 == len(want) and got == want
    finally:
        dist.destroy_process_group()


def test_in_library_collectives_world_size_one(vx):
    """vx355_comm_create / vx355_exchange_counts / vx355_exchange_columns / vx355_all_gather on a
    one-rank communicator (the only size a 1-GPU box offers): slices addressed to the own rank
    come back unchanged, counts and gathers are identities."""
    import torch
    dev = torch.device("cuda", 0)
    comm = vx.Comm(vx.Comm.unique_id(), 1, 0)
    assert comm.exchange_counts([12345]) == [12345]
    a = torch.arange(12345, dtype=torch.int64, device=dev) * 3
    b = torch.rand(12345, dtype=torch.float64, device=dev)
    c = torch.arange(12345 * 4, dtype=torch.int32, device=dev).reshape(12345, 4)
    outs = [torch.zeros_like(x) for x in (a, b, c)]
    torch.cuda.synchronize()
    comm.exchange_columns([x.data_ptr() for x in (a, b, c)], [8, 8, 16], [12345], [12345],
                          [o.data_ptr() for o in outs])
    for x, o in zip((a, b, c), outs):
        assert bool((x == o).all())
    g = torch.zeros_like(a)
    comm.all_gather(a.data_ptr(), g.data_ptr(), a.numel() * 8)
    assert bool((g == a).all())


@pytest.mark.parametrize("join_type", JOIN_TYPES)
def test_join_generic_hash_mode_keys(oracle, vx, join_type):
    """Join keys without a 64-bit normalized form — DOUBLE (NaN == NaN, 0.0 == -0.0
    as hashOne / the key compare treat them), 8..12-byte strings, three wide
    BIGINTs — run in the generic hash mode: VectorHasher hash, {tag, row} slots,
    stored key images; duplicates chained behind the first row of a key."""
    rng = np.random.default_rng(61)
    nb, npb = 5000, 16000
    dvals = np.array([0.0, -0.0, 1.5, float("nan"), -2.25, 1e300, -np.inf] + list(rng.random(300)))
    svals = [b"", b"A", b"BUILDING", b"AUTOMOBILE", b"HOUSEHOLD", b"twelve bytes", b"FURNITURE"]
    wide = rng.integers(-2 ** 62, 2 ** 62, 40).astype(np.int64)
    cases = {
        "double": ([abi.DOUBLE], lambda n: [dvals[rng.integers(0, len(dvals), n)]]),
        "string": ([abi.VARCHAR], lambda n: [[svals[i] for i in rng.integers(0, len(svals), n)]]),
        "wide": ([abi.BIGINT] * 3, lambda n: [wide[rng.integers(0, 40, n)], wide[rng.integers(0, 3, n)],
                                               wide[rng.integers(0, 2, n)]]),
        "mixed": ([abi.VARCHAR, abi.DOUBLE, abi.INTEGER],
                  lambda n: [[svals[i] for i in rng.integers(0, len(svals), n)], dvals[rng.integers(0, 7, n)],
                             rng.integers(0, 3, n).astype(np.int32)]),
    }
    for name, (kinds, gen) in cases.items():
        nk = len(kinds)
        bcols, pcols = gen(nb), gen(npb)
        bvalid = [rng.random(nb) > 0.05] + [None] * (nk - 1)
        pvalid = [rng.random(npb) > 0.05] + [None] * (nk - 1)
        pay = rng.integers(0, 1 << 40, nb).astype(np.int64)
        results = {}
        for impl in (oracle, vx):
            half = nb // 2

            def part(lo, hi):
                cols = [c[lo:hi] for c in bcols] + [pay[lo:hi]]
                valids = [None if v is None else v[lo:hi] for v in bvalid] + [None]
                return batch_of(cols, valids)
            table, _b = _build(impl, [[part(0, half)], [part(half, nb)]], list(range(nk)), kinds, [nk],
                               [abi.BIGINT], join_type)
            st = table.stats()
            probe = impl.JoinProbe(table, list(range(nk)), join_type)
            probe.add_input(batch_of(pcols, pvalid))
            pairs, payload = _drain(probe, 997)
            _contiguous(pairs)
            results[impl.__name__] = (_canon(pairs, payload), st.num_rows, st.num_distinct, st.has_duplicates)
            if impl is vx:
                assert st.hash_mode == abi.MODE_HASH, name
        assert results[oracle.__name__] == results[vx.__name__], name


def test_dynamic_filters_from_the_build_side(oracle, vx):
    """HashProbe::pushdownDynamicFilters (HashProbe.cpp:408-457): per join key either the distinct
    value list (<= VectorHasher::kMaxDistinct = 100 000 values -> createBigintValues) or Bloom
    blocks bit-identical to the reference's SplitBlockBloomFilter over folly::hasher<int64_t>."""
    from velox_amd import ops
    rng = np.random.default_rng(55)
    nb = 300000
    k0 = rng.integers(-2 ** 40, 2 ** 40, 200000)[rng.integers(0, 200000, nb)].astype(np.int64)  # > 100 K distinct
    k1 = rng.integers(-500, 700, nb).astype(np.int32)                                           # few distinct
    valid0 = rng.random(nb) > 0.02
    s = [[b"a", b"bb"][i] for i in rng.integers(0, 2, nb)]
    halves = [slice(0, nb // 2), slice(nb // 2, nb)]
    batches = [[batch_of([k0[h], k1[h], s[h]], [valid0[h], None, None])] for h in halves]
    table, builds = _build(vx, batches, [0, 1, 2], [abi.BIGINT, abi.INTEGER, abi.VARCHAR], [], [], abi.JOIN_INNER)
    kept0, kept1 = k0[valid0], k1[valid0]   # rows with a null key never reach the table
    # key 1: value list
    f1 = table.key_filter(1)
    assert f1.kind == abi.KEY_FILTER_VALUES and (f1.min, f1.max) == (kept1.min(), kept1.max())
    assert (table.key_filter_values(1) == np.unique(kept1)).all() and f1.num_distinct == len(np.unique(kept1))
    # key 0: Bloom blocks, both SIMD widths, equal to the reference algorithm's blocks
    f0 = table.key_filter(0)
    assert f0.kind == abi.KEY_FILTER_BLOOM and (f0.min, f0.max) == (kept0.min(), kept0.max())
    assert f0.num_distinct == table.stats().num_distinct
    for lanes in (8, 4):
        blocks = table.key_filter_bloom(0, lanes)
        assert blocks.shape[0] == oracle.bloom_num_blocks(f0.num_distinct, 0.01, lanes)
        exp = oracle.bloom_build(kept0, lanes, capacity=f0.num_distinct)
        assert (blocks == exp).all()
        # device-side test of a probe column: never a false negative, nulls and unselected rows fail
        probe = np.concatenate([kept0[:5000], rng.integers(-2 ** 40, 2 ** 40, 20000)]).astype(np.int64)
        pvalid = rng.random(len(probe)) > 0.1
        got = ops.bloom_test(blocks, abi.HostColumn(abi.BIGINT, probe, pvalid))
        assert (got == (oracle.bloom_test(exp, probe) & pvalid)).all() and got[:5000][pvalid[:5000]].all()
    # string keys: no filter (VectorHasher::getFilter for strings needs the opt-in config; not offered)
    assert table.key_filter(2).kind == abi.KEY_FILTER_NONE


def _drain_build_side(probe, max_rows):
    rows, payload = [], []
    while True:
        build_rows, cols, fin = probe.get_build_side_output(max_rows)
        assert len(build_rows) <= max_rows
        for i in range(len(build_rows)):
            rows.append(int(build_rows[i]))
            payload.append(tuple(None if not valid[i] else (vals[i] if isinstance(vals, list) else vals[i].item())
                                 for vals, valid in cols))
        if fin:
            break
    return rows, payload


@pytest.mark.parametrize("join_type", [abi.JOIN_RIGHT, abi.JOIN_FULL, abi.JOIN_RIGHT_SEMI_FILTER,
                                       abi.JOIN_LEFT_SEMI_PROJECT])
@pytest.mark.parametrize("mode", ["array", "normalized", "generic"])
def test_right_full_semi_project_joins(oracle, vx, join_type, mode, monkeypatch):
    """Right / full joins (unmatched build rows, null keys included, after the last probe), right
    semi filter (matched build rows once) and left semi project (every probe row + first match)
    in all three table modes, two build drivers, two probe batches on two probe handles."""
    if mode == "normalized":
        monkeypatch.setenv("VX355_JOIN_ARRAY_MAX", "0")
    rng = np.random.default_rng(31)
    nb, npb = 5000, 12000
    if mode == "generic":
        pool = rng.integers(-2 ** 62, 2 ** 62, 1500).astype(np.int64)
    else:
        pool = np.arange(1500, dtype=np.int64)
    bk = pool[rng.integers(0, 1500, nb)]
    bk2 = rng.integers(0, 3, nb).astype(np.int32)
    bvalid = rng.random(nb) > 0.05
    bpay = np.arange(nb, dtype=np.int64) * 7
    bpay_valid = rng.random(nb) > 0.2
    pk = np.concatenate([pool[rng.integers(0, 1200, npb - 2000)], rng.integers(-50, 0, 2000).astype(np.int64)])
    rng.shuffle(pk)
    pk2 = rng.integers(0, 3, npb).astype(np.int32)
    pvalid = rng.random(npb) > 0.05
    key_types = [abi.BIGINT, abi.INTEGER]
    results = {}
    for impl in (oracle, vx):
        def driver(lo, hi):
            return [abi.HostBatch([abi.HostColumn(abi.BIGINT, bk[lo:hi], bvalid[lo:hi]),
                                   abi.HostColumn(abi.INTEGER, bk2[lo:hi]),
                                   abi.HostColumn(abi.BIGINT, bpay[lo:hi], bpay_valid[lo:hi])])]
        table, builds = _build(impl, [driver(0, 3000), driver(3000, nb)], [0, 1], key_types, [2], [abi.BIGINT],
                               join_type)
        keeps_nulls = join_type in (abi.JOIN_RIGHT, abi.JOIN_FULL)
        assert table.stats().num_rows == (nb if keeps_nulls else bvalid.sum())
        if impl is vx:
            want = {"array": abi.MODE_ARRAY, "normalized": abi.MODE_NORMALIZED_KEY, "generic": abi.MODE_HASH}[mode]
            assert table.stats().hash_mode == want
        probes = [impl.JoinProbe(table, [0, 1], join_type) for _ in range(2)]
        out = []
        for h, sl in zip(probes, [slice(0, 7000), slice(7000, npb)]):
            h.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, pk[sl], pvalid[sl]),
                                       abi.HostColumn(abi.INTEGER, pk2[sl])]))
            pairs, payload = _drain(h, 1000 if impl is oracle else 513)
            _contiguous(pairs)
            out.append(_canon(pairs, payload))
        if join_type != abi.JOIN_LEFT_SEMI_PROJECT:
            out.append(_drain_build_side(probes[1], 999 if impl is oracle else 300))
        results[impl.__name__] = out
    if join_type == abi.JOIN_LEFT_SEMI_PROJECT:
        # The representative of a duplicate chain is not stable (parallel build): compare the match flag.
        flags = {k: [[(p[0], p[1] >= 0) for p, _ in part] for part in v] for k, v in results.items()}
        assert flags[oracle.__name__] == flags[vx.__name__]
        assert [p[0] for p, _ in results[vx.__name__][0]] == list(range(7000))
    else:
        assert results[oracle.__name__] == results[vx.__name__]
        assert len(results[vx.__name__][2][0]) > 0


@pytest.mark.parametrize("case", ["build_has_null", "regular", "empty_build", "only_null_build"])
@pytest.mark.parametrize("hash_mode", [False, True])
def test_null_aware_left_semi_project(oracle, vx, case, hash_mode):
    """x IN (subquery) as a column (HashProbe::fillLeftSemiProjectMatchColumn, HashProbe.cpp:923-966):
    TRUE / FALSE / NULL per probe row; build_rows_out carries first match / -1 / -2."""
    rng = np.random.default_rng(33)
    nb, npb = 3000, 9000
    spread = 10 ** 12 if hash_mode else 800
    bk = (rng.integers(0, 800, nb) * (spread // 800)).astype(np.int64)
    bvalid = rng.random(nb) > 0.02 if case == "build_has_null" else np.ones(nb, dtype=bool)
    if case == "empty_build":
        bk, bvalid = bk[:0], bvalid[:0]
    if case == "only_null_build":
        bk, bvalid = bk[:5], np.zeros(5, dtype=bool)
    pk = (rng.integers(-100, 1200, npb) * (spread // 800)).astype(np.int64)
    pvalid = rng.random(npb) > 0.05
    got = {}
    for impl in (oracle, vx):
        b = impl.JoinBuild([0], [abi.BIGINT], [], [], abi.JOIN_LEFT_SEMI_PROJECT, True)
        b.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, bk, bvalid)], len(bk)))
        table = b.finish()
        probe = impl.JoinProbe(table, [0], abi.JOIN_LEFT_SEMI_PROJECT, True)
        probe.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, pk, pvalid)]))
        rows_out = []
        while True:
            mapping, rows, _, fin = probe.get_output(1000, [])
            rows_out += [(int(m), "T" if r >= 0 else ("F" if r == -1 else "N")) for m, r in zip(mapping, rows)]
            if fin:
                break
        got[impl.__name__] = rows_out
    assert got[oracle.__name__] == got[vx.__name__]
    keys = set(bk[bvalid].tolist())
    has_null = bool((~bvalid).any())
    for i, (m, flag) in enumerate(got[vx.__name__]):
        assert m == i
        if not keys:
            want = "N" if has_null else "F"
        elif not pvalid[i]:
            want = "N"
        elif int(pk[i]) in keys:
            want = "T"
        else:
            want = "N" if has_null else "F"
        assert flag == want, (i, flag, want)


@pytest.mark.parametrize("case", ["build_has_null", "regular", "empty_build"])
def test_null_aware_anti_join(oracle, vx, case):
    """NOT IN (HashProbe.cpp:1316-1328 + HashBuild's antiJoinHasNullKeys)."""
    rng = np.random.default_rng(32)
    nb, npb = 3000, 9000
    bk = rng.integers(0, 800, nb).astype(np.int64)
    bvalid = rng.random(nb) > 0.02 if case == "build_has_null" else np.ones(nb, dtype=bool)
    if case == "empty_build":
        bk, bvalid = bk[:0], bvalid[:0]
    pk = rng.integers(-100, 1200, npb).astype(np.int64)
    pvalid = rng.random(npb) > 0.05
    got = {}
    for impl in (oracle, vx):
        b = impl.JoinBuild([0], [abi.BIGINT], [], [], abi.JOIN_ANTI, True)
        b.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, bk, bvalid)], len(bk)))
        table = b.finish()
        probe = impl.JoinProbe(table, [0], abi.JOIN_ANTI, True)
        probe.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, pk, pvalid)]))
        pairs, _ = _drain(probe, 4096)
        got[impl.__name__] = [p[0] for p in pairs]
    assert got[oracle.__name__] == got[vx.__name__]
    if case == "build_has_null":
        assert got[vx.__name__] == []
    elif case == "empty_build":
        assert got[vx.__name__] == list(range(npb))
    else:
        keys = set(bk.tolist())
        assert got[vx.__name__] == [i for i in range(npb) if pvalid[i] and int(pk[i]) not in keys]


@pytest.mark.parametrize("sparse", ["0", "1", "adaptive"])
@pytest.mark.parametrize("join_type", [abi.JOIN_INNER, abi.JOIN_LEFT_SEMI_FILTER, abi.JOIN_RIGHT])
@pytest.mark.parametrize("force_hash", [False, True])
def test_low_hit_rate_joins_list_their_hits_in_the_probe_pass(oracle, vx, sparse, join_type, force_hash, monkeypatch):
    """Joins that emit matches only over unique build keys: the probe pass lists the hits of every
    8192-row tile itself (k_join_probe_list, listJoinResultsFastPath's role) instead of writing
    hits[] for every probe row. The probe rows below mix long runs of misses, a stretch where every
    row hits (tiles that overflow their staging segment fall back to the dense form) and the tail
    of the batch; outputs are drained in windows that cut through tiles. 3 M probe rows = 367 tiles:
    the adaptive mode measures the first 256 and decides for the rest."""
    if sparse != "adaptive":
        monkeypatch.setenv("VX355_JOIN_SPARSE", sparse)
    if force_hash:
        monkeypatch.setenv("VX355_JOIN_ARRAY_MAX", "0")
    rng = np.random.default_rng(31)
    nb, npb = 40000, 3_000_000
    bk = (rng.permutation(4_000_000)[:nb]).astype(np.int64)
    pay = rng.integers(0, 1 << 30, nb).astype(np.int64)
    pk = rng.integers(0, 4_000_000, npb).astype(np.int64)          # ~1 % hit
    pk[1_000_000:1_030_000] = bk[rng.integers(0, nb, 30000)]       # a dense stretch: every row hits
    pk[-5:] = bk[:5]
    res = {}
    for impl in (oracle, vx):
        table, _ = _build(impl, [[batch_of([bk, pay])]], [0], [abi.BIGINT], [1], [abi.BIGINT], join_type)
        probe = impl.JoinProbe(table, [0], join_type)
        if impl is vx:
            vx.profile_reset()
            vx.profile_enable(True)
        probe.add_input(batch_of([pk]))
        maps, rows, pays = [], [], []
        while True:
            m, r, cols, fin = probe.get_output(7001, [0])
            maps.append(np.asarray(m))
            rows.append(np.asarray(r))
            pays.append(np.where(np.asarray(cols[0][1]), np.asarray(cols[0][0]), -1))
            if fin:
                break
        if impl is vx:
            vx.profile_enable(False)
            names = set(vx.profile().keys())
            assert ("k_join_probe_list" in names) == (sparse != "0")
        res[impl.__name__] = (np.concatenate(maps), np.concatenate(rows), np.concatenate(pays))
        if join_type == abi.JOIN_RIGHT:
            bm, br, bcols, bfin = [], [], [], False
            while not bfin:
                r2, c2, bfin = probe.get_build_side_output(9000, [0])
                br.append(np.asarray(r2))
                bcols.append(np.asarray(c2[0][0]))
            res[impl.__name__] += (np.concatenate(br), np.concatenate(bcols))
    for g, e in zip(res[vx.__name__], res[oracle.__name__]):
        assert len(g) == len(e) and (g == e).all()
    assert len(res[vx.__name__][0]) > 30000


ALL_FILTER_KINDS = [abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_RIGHT, abi.JOIN_FULL, abi.JOIN_LEFT_SEMI_FILTER,
                    abi.JOIN_LEFT_SEMI_PROJECT, abi.JOIN_ANTI, abi.JOIN_RIGHT_SEMI_FILTER,
                    abi.JOIN_RIGHT_SEMI_PROJECT, abi.JOIN_RIGHT_ANTI]


def _probe_all(impl, table, key_cols, join_type, batches, flt, max_rows, dep_ids):
    """-> (sorted pairs per batch incl. payload, build-side output) with the oracle's conventions."""
    probe = impl.JoinProbe(table, key_cols, join_type)
    if flt:
        probe.set_filter(flt)
    out = []
    for hb in batches:
        probe.add_input(hb)
        pairs, payload = _drain(probe, max_rows, dep_ids)
        _contiguous(pairs)
        if join_type == abi.JOIN_LEFT_SEMI_PROJECT:
            pairs = [(r, 0 if b >= 0 else -1) for r, b in pairs]    # which match is reported is chain order
            payload = [() for _ in payload]
        out.append(_canon(pairs, payload))
    side = None
    if join_type in (abi.JOIN_RIGHT, abi.JOIN_FULL, abi.JOIN_RIGHT_ANTI, abi.JOIN_RIGHT_SEMI_FILTER,
                     abi.JOIN_RIGHT_SEMI_PROJECT):
        ids = list(dep_ids) + ([abi.BUILD_COL_MATCH] if join_type == abi.JOIN_RIGHT_SEMI_PROJECT else [])
        rows, cols_acc = [], None
        while True:
            r, cols, fin = probe.get_build_side_output(max_rows, ids)
            rows += r.tolist()
            vals = [[None if not ok else (v.item() if hasattr(v, "item") else v) for v, ok in zip(c[0], c[1])]
                    for c in cols]
            cols_acc = vals if cols_acc is None else [a + b for a, b in zip(cols_acc, vals)]
            if fin:
                break
        side = (rows, cols_acc)
    return out, side


@pytest.mark.parametrize("join_type", ALL_FILTER_KINDS)
@pytest.mark.parametrize("flt", [None, "int_lt", "mixed"])
@pytest.mark.parametrize("force_hash", [False, True])
def test_every_join_kind_with_and_without_extra_filter(oracle, vx, join_type, flt, force_hash, monkeypatch):
    """HashProbe::evalFilter semantics per join kind (HashProbe.cpp:1487-1713) and the build-side
    outputs of the right-side kinds incl. right semi project's match column and right anti, vs the
    oracle (itself pinned to nested loops in test_oracle_ops.py). Duplicate build keys, null keys
    and null filter operands on both sides; two probe batches share the probed flags."""
    if force_hash:
        monkeypatch.setenv("VX355_JOIN_ARRAY_MAX", "0")
    rng = np.random.default_rng(200 + join_type)
    nb, npb = 5000, 30000
    bk = rng.integers(0, 1200, nb).astype(np.int64)
    bvalid = rng.random(nb) > 0.05
    bw = rng.integers(0, 100, nb).astype(np.int32)
    bwvalid = rng.random(nb) > 0.1
    bd = rng.random(nb) * 100
    bs = [bytes(rng.choice([b"AIR", b"SHIP", b"TRUCK", b"REG AIR", b"twelve bytes"])) for _ in range(nb)]
    pk = rng.integers(-50, 1500, npb).astype(np.int64)
    pvalid = rng.random(npb) > 0.05
    pv = rng.integers(0, 100, npb).astype(np.int64)
    pvvalid = rng.random(npb) > 0.1
    ps = [bytes(rng.choice([b"AIR", b"SHIP", b"TRUCK", b"REG AIR", b"twelve bytes"])) for _ in range(npb)]
    terms = {None: None,
             "int_lt": [(("probe", 1), abi.CMP_LT, ("build", 0))],
             "mixed": [(("build", 1), abi.CMP_GE, ("probe", 1)),            # DOUBLE vs BIGINT: compares as double
                       (("probe", 2), abi.CMP_EQ, ("build", 2)),            # inline strings
                       (("build", 0), abi.CMP_NE, 17), (("probe", 2), abi.CMP_NE, b"TRUCK")]}[flt]
    build_batch = abi.HostBatch([abi.HostColumn(abi.BIGINT, bk, bvalid), abi.HostColumn(abi.INTEGER, bw, bwvalid),
                                 abi.HostColumn(abi.DOUBLE, bd), abi.HostColumn(abi.VARCHAR, bs)])
    half = npb // 2
    probe_batches = [abi.HostBatch([abi.HostColumn(abi.BIGINT, pk[lo:hi], pvalid[lo:hi]),
                                    abi.HostColumn(abi.BIGINT, pv[lo:hi], pvvalid[lo:hi]),
                                    abi.HostColumn(abi.VARCHAR, ps[lo:hi])])
                     for lo, hi in ((0, half), (half, npb))]
    res = {}
    for impl in (oracle, vx):
        table, _ = _build(impl, [[build_batch]], [0], [abi.BIGINT], [1, 2, 3], [abi.INTEGER, abi.DOUBLE, abi.VARCHAR],
                          join_type)
        res[impl.__name__] = _probe_all(impl, table, [0], join_type, probe_batches, terms,
                                        997 if impl is vx else 1500, [0, 2])
    assert res[vx.__name__][0] == res[oracle.__name__][0]
    assert res[vx.__name__][1] == res[oracle.__name__][1]
    if join_type in (abi.JOIN_INNER, abi.JOIN_LEFT):
        assert sum(len(b) for b in res[vx.__name__][0]) > 100


@pytest.mark.parametrize("join_type", [abi.JOIN_COUNTING_LEFT_SEMI_FILTER, abi.JOIN_COUNTING_ANTI])
@pytest.mark.parametrize("force_hash", [False, True])
def test_counting_joins_intersect_all_except_all(oracle, vx, join_type, force_hash, monkeypatch):
    """core/PlanNode.h:3112-3116,3152-3156, HashProbe.cpp:1345-1365: every match consumes one
    occurrence of the build key, in probe-row order, across batches; two key columns."""
    if force_hash:
        monkeypatch.setenv("VX355_JOIN_ARRAY_MAX", "0")
    rng = np.random.default_rng(77)
    nb, npb = 20000, 60000
    bk1 = rng.integers(0, 300, nb).astype(np.int64)
    bk2 = rng.integers(0, 20, nb).astype(np.int32)
    bvalid = rng.random(nb) > 0.02
    pk1 = rng.integers(-10, 320, npb).astype(np.int64)
    pk2 = rng.integers(0, 22, npb).astype(np.int32)
    pvalid = rng.random(npb) > 0.02
    res = {}
    for impl in (oracle, vx):
        table, _ = _build(impl, [[abi.HostBatch([abi.HostColumn(abi.BIGINT, bk1, bvalid),
                                                 abi.HostColumn(abi.INTEGER, bk2)])]],
                          [0, 1], [abi.BIGINT, abi.INTEGER], [], [], join_type)
        probe = impl.JoinProbe(table, [0, 1], join_type)
        got = []
        for lo in range(0, npb, 25000):
            hi = min(npb, lo + 25000)
            probe.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, pk1[lo:hi], pvalid[lo:hi]),
                                           abi.HostColumn(abi.INTEGER, pk2[lo:hi])]))
            while True:
                m, r, cols, fin = probe.get_output(7777, [])
                got.append(np.asarray(m) + lo)
                if fin:
                    break
        res[impl.__name__] = np.concatenate(got)
    assert len(res[vx.__name__]) == len(res[oracle.__name__]) and (res[vx.__name__] == res[oracle.__name__]).all()
    assert 1000 < len(res[vx.__name__]) < npb - 1000


def test_unsupported_join_flavours_are_refused_at_create(vx):
    b = vx.JoinBuild([0], [abi.BIGINT], [], [], abi.JOIN_COUNTING_ANTI)
    b.add_input(batch_of([np.arange(10, dtype=np.int64)]))
    t = b.finish()
    p = vx.JoinProbe(t, [0], abi.JOIN_COUNTING_ANTI)
    with pytest.raises(vx.Vx355Error) as e:
        p.set_filter([(("probe", 0), abi.CMP_LT, 5)])
    assert e.value.status == abi.EUNSUPPORTED
    with pytest.raises(vx.Vx355Error):
        vx.JoinProbe(t, [0], abi.JOIN_INNER)               # counting table, non-counting probe
    with pytest.raises(vx.Vx355Error) as e:                # null aware exists for ANTI and LEFT_SEMI_PROJECT
        vx.JoinBuild([0], [abi.BIGINT], [], [], abi.JOIN_RIGHT_SEMI_PROJECT, null_aware=True)
    assert e.value.status == abi.EUNSUPPORTED


def _join_arrays(impl, vxmod, bk, pay, pk, join_type, max_rows=500000):
    table, _ = _build(impl, [[batch_of([bk, pay])]], [0], [abi.BIGINT], [1], [abi.BIGINT], join_type)
    probe = impl.JoinProbe(table, [0], join_type)
    if impl is vxmod:
        vxmod.profile_reset()
        vxmod.profile_enable(True)
    probe.add_input(batch_of([pk]))
    maps, rows, pays = [], [], []
    while True:
        m, r, cols, fin = probe.get_output(max_rows, [0])
        maps.append(np.asarray(m))
        rows.append(np.asarray(r))
        pays.append(np.where(np.asarray(cols[0][1]), np.asarray(cols[0][0]), -1))
        if fin:
            break
    names = set()
    if impl is vxmod:
        vxmod.profile_enable(False)
        names = set(vxmod.profile().keys())
    out = [np.concatenate(maps), np.concatenate(rows), np.concatenate(pays)]
    if join_type == abi.JOIN_RIGHT:
        br, bfin = [], False
        while not bfin:
            r2, c2, bfin = probe.get_build_side_output(max_rows, [0])
            br.append(np.asarray(r2))
        out.append(np.concatenate(br))
    return out, names


@pytest.mark.parametrize("join_type", [abi.JOIN_INNER, abi.JOIN_LEFT_SEMI_FILTER, abi.JOIN_RIGHT])
def test_range_partitioned_probe_forced(oracle, vx, join_type, monkeypatch):
    """k_pp_count / k_pp_scatter / k_join_probe_part: probe rows range-partitioned by key, every bin
    probed against its slice of the presence bitmap in LDS, hits re-ordered by probe row. Forced on
    (VX355_JOIN_PARTITION=1) for a key range of 8 bins; keys below / above the build range, a
    stretch of certain hits and windows that cut through the output."""
    monkeypatch.setenv("VX355_JOIN_PARTITION", "1")
    rng = np.random.default_rng(91)
    nb, npb = 70000, 1_500_000      # 64 x build rows >= the key range: the table stays in array mode
    bk = (rng.permutation(4_000_000)[:nb] + 1000).astype(np.int64)
    pay = rng.integers(0, 1 << 30, nb).astype(np.int64)
    pk = rng.integers(0, 4_100_000, npb).astype(np.int64)
    pk[700_000:705_000] = bk[rng.integers(0, nb, 5000)]
    res = {}
    for impl in (oracle, vx):
        res[impl.__name__], names = _join_arrays(impl, vx, bk, pay, pk, join_type, max_rows=7001)
        if impl is vx:
            assert "k_join_probe_part" in names and "k_pp_scatter" in names
    for g, e in zip(res[vx.__name__], res[oracle.__name__]):
        assert len(g) == len(e) and (g == e).all()
    assert len(res[vx.__name__][0]) > 15000


@pytest.mark.parametrize("order", ["random", "clustered"])
def test_range_partitioned_probe_is_chosen_for_scattered_keys_only(oracle, vx, order):
    """Adaptive choice: a 25 MB presence bitmap (200 M possible keys), 5 M probe rows. Random probe
    order takes the partitioned path, key-ordered probe rows (what a fact table stored in key order
    gives) keep the direct one. Results equal the oracle's either way."""
    rng = np.random.default_rng(92)
    nb, npb = 3_300_000, 5_000_000
    bk = np.unique(rng.integers(0, 200_000_000, nb + nb // 8))     # (a 200 M-element permutation takes minutes)
    bk = rng.permutation(bk)[:nb].astype(np.int64)
    pay = np.arange(nb, dtype=np.int64)
    pk = rng.integers(0, 200_000_000, npb).astype(np.int64)
    pk[::50] = bk[rng.integers(0, nb, len(pk[::50]))]
    if order == "clustered":
        pk = np.sort(pk)
    res = {}
    for impl in (oracle, vx):
        res[impl.__name__], names = _join_arrays(impl, vx, bk, pay, pk, abi.JOIN_INNER)
        if impl is vx:
            assert ("k_join_probe_part" in names) == (order == "random"), names
    for g, e in zip(res[vx.__name__], res[oracle.__name__]):
        assert len(g) == len(e) and (g == e).all()
    assert len(res[vx.__name__][0]) >= npb // 50


def test_output_pages_respect_preferred_output_batch_bytes(oracle, vx):
    """listJoinResults' byte bound (HashTable.cpp:2087-2153, QueryConfig::preferredOutputBatchBytes):
    pages stop at the byte budget of the projected build columns; the concatenation is unchanged."""
    rng = np.random.default_rng(95)
    bk = rng.integers(0, 3000, 20000).astype(np.int64)
    pay = rng.integers(0, 1 << 40, 20000).astype(np.int64)
    pay2 = rng.integers(0, 1 << 20, 20000).astype(np.int32)
    pk = rng.integers(0, 4000, 30000).astype(np.int64)
    res = {}
    for impl in (oracle, vx):
        table, _ = _build(impl, [[batch_of([bk, pay, pay2])]], [0], [abi.BIGINT], [1, 2], [abi.BIGINT, abi.INTEGER],
                          abi.JOIN_INNER)
        probe = impl.JoinProbe(table, [0], abi.JOIN_INNER)
        if impl is vx:
            probe.set_output_batch_bytes(12 * 1000)          # 12 bytes per row -> at most 1000 rows per page
        probe.add_input(batch_of([pk]))
        pages, pairs, payload = [], [], []
        while True:
            m, r, cols, fin = probe.get_output(50000, [0, 1])
            pages.append(len(m))
            pairs += list(zip(m.tolist(), r.tolist()))
            if fin:
                break
        res[impl.__name__] = sorted(pairs)
        if impl is vx:
            assert max(pages) == 1000 and sum(pages) == len(pairs) and len(pages) > 100
    assert res[vx.__name__] == res[oracle.__name__]


@pytest.mark.parametrize("join_type", [abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_RIGHT, abi.JOIN_ANTI])
def test_join_keys_and_payloads_longer_than_12_bytes(oracle, vx, join_type):
    """Non-inline StringViews (size > 12) as join keys and as payload: the build side copies them into
    its HBM arena, the generic hash mode hashes and compares them by content, gathered payload views
    reach host outputs through a buffer owned by the probe handle. Strings around the inline / prefix
    boundaries, pairs that differ only past the prefix, nulls, two build drivers, an extra filter on
    the long payload (=)."""
    rng = np.random.default_rng(97)
    lens = [0, 3, 12, 13, 16, 17, 24, 33, 64, 200]
    words = list(dict.fromkeys((b"%05d/" % i + bytes(rng.integers(97, 123, 250).astype(np.uint8)))[:lens[i % len(lens)]]
                               for i in range(300)))
    words += [b"same-prefix-then-differs-A", b"same-prefix-then-differs-B"]
    nb, npb = 4000, 15000
    bk = [words[i] for i in rng.integers(0, len(words), nb)]
    bvalid = rng.random(nb) > 0.05
    bpay = [words[i] for i in rng.integers(0, len(words), nb)]
    bpvalid = rng.random(nb) > 0.1
    bnum = rng.integers(0, 100, nb).astype(np.int64)
    pk = [words[i] for i in rng.integers(0, len(words), npb)] 
    pvalid = rng.random(npb) > 0.05
    ps = [words[i] for i in rng.integers(0, len(words), npb)]
    res = {}
    for with_filter in (False, True):
        for impl in (oracle, vx):
            half = nb // 2
            parts = [[abi.HostBatch([abi.HostColumn(abi.VARCHAR, bk[lo:hi], bvalid[lo:hi]),
                                     abi.HostColumn(abi.VARCHAR, bpay[lo:hi], bpvalid[lo:hi]),
                                     abi.HostColumn(abi.BIGINT, bnum[lo:hi])])] for lo, hi in ((0, half), (half, nb))]
            table, _ = _build(impl, parts, [0], [abi.VARCHAR], [1, 2], [abi.VARCHAR, abi.BIGINT], join_type)
            probe = impl.JoinProbe(table, [0], join_type)
            if with_filter:
                probe.set_filter([(("build", 1), abi.CMP_LT, 60)])
            probe.add_input(abi.HostBatch([abi.HostColumn(abi.VARCHAR, pk, pvalid), abi.HostColumn(abi.VARCHAR, ps)]))
            pairs, payload = _drain(probe, 1013)
            _contiguous(pairs)
            out = [_canon(pairs, payload)]
            if join_type == abi.JOIN_RIGHT:
                rows, strs = [], []
                while True:
                    r, cols, fin = probe.get_build_side_output(777, [0])
                    rows += r.tolist()
                    strs += [v if ok else None for v, ok in zip(cols[0][0], cols[0][1])]
                    if fin:
                        break
                out.append((rows, strs))
            res[(impl.__name__, with_filter)] = out
            if impl is vx:
                assert table.stats().hash_mode == abi.MODE_HASH
        assert res[(vx.__name__, with_filter)] == res[(oracle.__name__, with_filter)]
    if join_type == abi.JOIN_INNER:
        assert any(p[0] is not None and len(p[0]) > 12 for _, p in res[(vx.__name__, False)][0])


@pytest.mark.parametrize("shape", ["array", "normalized", "hash", "two_keys", "string_key"])
@pytest.mark.parametrize("join_type", [abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_FULL, abi.JOIN_LEFT_SEMI_FILTER, abi.JOIN_ANTI,
                                       abi.JOIN_COUNTING_LEFT_SEMI_FILTER, abi.JOIN_COUNTING_ANTI])
def test_null_as_value_joins(oracle, vx, shape, join_type):
    """HashJoinNode::isNullAsValue (IS NOT DISTINCT FROM keys; what INTERSECT / EXCEPT plan their
    counting joins with, core/PlanNode.h:3442-3445): NULL keys are values on both sides, in every
    table mode; oracle parity (the oracle itself is checked against nested loops on the CPU)."""
    counting = join_type in (abi.JOIN_COUNTING_LEFT_SEMI_FILTER, abi.JOIN_COUNTING_ANTI)
    rng = np.random.default_rng(1234 + hash(shape) % 100)
    nb, npb = 4000, 15000
    if shape == "array":
        kinds, spread = [abi.BIGINT], [1]
    elif shape == "normalized":
        kinds, spread = [abi.BIGINT], [10 ** 12]
    elif shape == "hash":
        kinds, spread = [abi.DOUBLE], [1]
    elif shape == "two_keys":
        kinds, spread = [abi.INTEGER, abi.BIGINT], [1, 1]
    else:
        kinds, spread = [abi.VARCHAR, abi.BIGINT], [1, 1]

    def column(kind, n, mult):
        if kind == abi.VARCHAR:
            words = [b"", b"a", b"bb", b"a key of more than twelve bytes", b"a key of more than twelve byteS"]
            return [words[i] for i in rng.integers(0, len(words), n)]
        v = rng.integers(0, 40, n) * mult
        return v.astype({abi.BIGINT: np.int64, abi.INTEGER: np.int32, abi.DOUBLE: np.float64}[kind])

    nk = len(kinds)
    bcols = [column(kinds[k], nb, spread[k]) for k in range(nk)]
    pcols = [column(kinds[k], npb, spread[k]) for k in range(nk)]
    bvalids = [rng.random(nb) > 0.1 for _ in kinds]
    pvalids = [rng.random(npb) > 0.1 for _ in kinds]
    results = {}
    for impl in (oracle, vx):
        deps = ([], []) if counting else ([nk], [abi.BIGINT])
        halves = []
        for lo, hi in ((0, nb // 2), (nb // 2, nb)):
            b = impl.JoinBuild(list(range(nk)), kinds, deps[0], deps[1], join_type, False, True)
            cols = [abi.HostColumn(kinds[k], bcols[k][lo:hi], bvalids[k][lo:hi]) for k in range(nk)]
            cols.append(abi.HostColumn(abi.BIGINT, np.arange(lo, hi, dtype=np.int64)))
            b.add_input(abi.HostBatch(cols, hi - lo))
            halves.append(b)
        table = halves[0].finish(halves[1:])
        probe = impl.JoinProbe(table, list(range(nk)), join_type, False, True)
        probe.add_input(abi.HostBatch([abi.HostColumn(kinds[k], pcols[k], pvalids[k]) for k in range(nk)], npb))
        rows_out = []
        while True:
            mapping, rows, cols, fin = probe.get_output(1777)
            pay = cols[0] if cols else None
            for i, m in enumerate(mapping):
                rows_out.append((int(m), None if pay is None or not pay[1][i] else int(pay[0][i])))
            if fin:
                break
        build_side = []
        if join_type == abi.JOIN_FULL:
            while True:
                rows, cols, fin = probe.get_build_side_output(999)
                build_side += [int(cols[0][0][i]) for i in range(len(rows))]
                if fin:
                    break
        # matches of one probe row are compared as a set
        results[impl.__name__] = (sorted(rows_out, key=lambda t: (t[0], -1 if t[1] is None else t[1])), sorted(build_side))
    assert results[oracle.__name__] == results[vx.__name__]
    # some probe rows with a null key found a build row with a null key
    if join_type == abi.JOIN_INNER:
        null_probe = {i for i in range(npb) if not all(pvalids[k][i] for k in range(nk))}
        assert any(m in null_probe for m, _ in results[vx.__name__][0])


def test_null_as_value_excludes_null_aware_and_must_match_the_table(vx):
    with pytest.raises(vx.Vx355Error) as e:
        vx.JoinBuild([0], [abi.BIGINT], [], [], abi.JOIN_ANTI, True, True)
    assert e.value.status == abi.EINVAL
    t = vx.JoinBuild([0], [abi.BIGINT], [], [], abi.JOIN_INNER, False, True).finish()
    with pytest.raises(vx.Vx355Error) as e:
        vx.JoinProbe(t, [0], abi.JOIN_INNER)
    assert e.value.status == abi.EINVAL


@pytest.mark.parametrize("build_nulls", [0.0, 0.1, 1.0])
@pytest.mark.parametrize("join_type", [abi.JOIN_ANTI, abi.JOIN_LEFT_SEMI_PROJECT])
def test_null_aware_joins_with_extra_filter(oracle, vx, join_type, build_nulls):
    """x [NOT] IN (SELECT y FROM build WHERE probe.v < build.w), HashProbe::evalFilterForNullAwareJoin
    (HashProbe.cpp:1639-1700): rows without a passing equal-key pair meet the null-key build rows
    (key not null) or every build row (key null). GPU == oracle == SQL's three-valued logic, two
    build drivers, output drained in small pages; larger case GPU == oracle."""
    from test_oracle_ops import null_aware_filter_case, run_null_aware_filter_join, sql_in_with_filter
    case = null_aware_filter_case(91 + join_type, build_nulls=build_nulls)
    bk, bvalid, bw, bwvalid, pk, pvalid, pv, pvvalid = case
    truth = sql_in_with_filter(pk, pvalid, pv, pvvalid, bk, bvalid, bw, bwvalid)
    got = run_null_aware_filter_join(vx, join_type, case)
    exp = run_null_aware_filter_join(oracle, join_type, case)
    if join_type == abi.JOIN_ANTI:
        assert got == exp and [i for i, _ in got] == [i for i, t in enumerate(truth) if t is False]
    else:
        state = lambda r: True if r >= 0 else (None if r == -2 else False)   # which match is reported is chain order
        assert [(i, state(r)) for i, r in got] == [(i, state(r)) for i, r in exp] == list(enumerate(truth))
    big = null_aware_filter_case(191 + join_type, nb=3000, npb=60000, build_nulls=build_nulls)
    got = run_null_aware_filter_join(vx, join_type, big, max_rows=7777)
    exp = run_null_aware_filter_join(oracle, join_type, big, max_rows=7777)
    if join_type == abi.JOIN_ANTI:
        assert got == exp
    else:
        assert [(i, min(r, 0)) for i, r in got] == [(i, min(r, 0)) for i, r in exp]


@pytest.mark.parametrize("join_type", [abi.JOIN_LEFT_SEMI_FILTER, abi.JOIN_LEFT_SEMI_PROJECT, abi.JOIN_ANTI])
@pytest.mark.parametrize("mode", ["array", "normalized", "generic"])
def test_build_side_drops_duplicate_keys_for_semi_and_anti_joins(oracle, vx, join_type, mode, monkeypatch):
    """HashJoinNode::canDropDuplicates (core/PlanNode.h:3391-3398; HashBuild.cpp:517-548): a left
    semi (filter / project) or anti join without an extra filter only asks whether a key exists, so
    the build links one row per key. Same probe rows out as the oracle's join over the full build
    side, the table reports unique keys, and what does not fit the rule is refused."""
    if mode == "normalized":
        monkeypatch.setenv("VX355_JOIN_ARRAY_MAX", "0")
    rng = np.random.default_rng(77 + join_type)
    nb, npb = 30000, 50000
    if mode == "generic":
        bk = [b"key-%05d-of-the-build-side" % k for k in rng.integers(0, 2000, nb)]
        pk = [b"key-%05d-of-the-build-side" % k for k in rng.integers(-50, 4000, npb)]
        kind = abi.VARCHAR
    else:
        bk = rng.integers(0, 2000, nb).astype(np.int64)
        pk = rng.integers(-50, 4000, npb).astype(np.int64)
        kind = abi.BIGINT
    bvalid = rng.random(nb) > 0.03
    pvalid = rng.random(npb) > 0.03
    build = abi.HostBatch([abi.HostColumn(kind, bk, bvalid)])
    probe_batch = abi.HostBatch([abi.HostColumn(kind, pk, pvalid)])
    results = {}
    for impl in (oracle, vx):
        kw = {"drop_duplicates": True} if impl is vx else {}
        b = impl.JoinBuild([0], [kind], [], [], join_type, **kw)
        for lo in range(0, nb, 7000):
            cols = [abi.HostColumn(kind, bk[lo:lo + 7000], bvalid[lo:lo + 7000])]
            b.add_input(abi.HostBatch(cols))
        table = b.finish()
        st = table.stats()
        if impl is vx:
            assert st.has_duplicates == 0
            assert st.num_distinct == len(set(k for k, v in zip(bk, bvalid) if v))
        p = impl.JoinProbe(table, [0], join_type)
        p.add_input(probe_batch)
        pairs, _ = _drain(p, 4096)
        # the first match's row number may differ (any row of the key stands for it): compare the
        # probe rows and, for the semi project, whether each of them matched
        results[impl.__name__] = [(r, b_row >= 0 if join_type == abi.JOIN_LEFT_SEMI_PROJECT else True) for r, b_row in pairs]
        if impl is vx:
            with pytest.raises(vx.Vx355Error) as e:
                p2 = vx.JoinProbe(table, [0], join_type)
                p2.set_filter([(("probe", 0), abi.CMP_GE, 0)])
            assert e.value.status == abi.EINVAL
    assert results[oracle.__name__] == results[vx.__name__] and len(results[vx.__name__]) > 1000
    del build
    with pytest.raises(vx.Vx355Error) as e:
        vx.JoinBuild([0], [kind], [], [], abi.JOIN_INNER, drop_duplicates=True)
    assert e.value.status == abi.EINVAL


@pytest.mark.parametrize("join_type", [abi.JOIN_INNER, abi.JOIN_LEFT_SEMI_FILTER, abi.JOIN_RIGHT])
@pytest.mark.parametrize("shape", ["array_listing", "array_dense_hits", "array_partitioned", "normalized", "hash",
                                   "dictionary_key", "two_keys_nullable_filter_column", "array_bigint_column_ne",
                                   "array_partitioned_le", "normalized_wide_slots_requested"])
def test_input_filter_fused_into_the_probe(oracle, vx, shape, join_type, monkeypatch):
    """FilterProject -> HashProbe fusion (vx355_join_probe_set_input_filter): probing the UNFILTERED
    batch with the filter inside the probe kernels equals filtering first (numpy here, FilterProject
    in a plan) and probing the selected rows with the oracle, the mappings composed with the
    selection (exec/OperatorUtils.cpp:393-422). Every probe kernel family: the listing probe, the
    dense form (tiles full of hits), the range-partitioned probe (both scatter variants), normalized
    keys, generic hash mode, a dictionary-wrapped key, two keys with a nullable filter column and a
    second term on a DOUBLE column."""
    rng = np.random.default_rng(123)
    nb, npb = 70000, 700_000      # 64 x build rows >= the key range: array mode unless forced otherwise
    space = 4_000_000
    if shape.startswith("array_partitioned"):
        monkeypatch.setenv("VX355_JOIN_PARTITION", "1")
    if shape.startswith("normalized"):
        monkeypatch.setenv("VX355_JOIN_ARRAY_MAX", "0")
        space = 1 << 40
    if shape == "normalized_wide_slots_requested":
        # the inline-dependent ("wide") probe instantiations carry no input filter: with one they must not be chosen
        monkeypatch.setenv("VX355_JOIN_WIDE", "1")
    bk = np.unique(rng.integers(0, space, nb + nb // 4)).astype(np.int64)
    bk = rng.permutation(bk)[:nb]
    pay = rng.integers(0, 1 << 30, nb).astype(np.int64)
    pk = rng.integers(0, space, npb).astype(np.int64)
    pk[::40] = bk[rng.integers(0, nb, len(pk[::40]))]
    if shape == "array_dense_hits":
        pk[100_000:200_000] = bk[rng.integers(0, nb, 100_000)]
    date = rng.integers(9000, 9400, npb).astype(np.int32)
    price = rng.integers(0, 1000, npb).astype(np.float64) / 8
    valid_date = None
    terms = [(1, abi.CMP_GT, 9204)]
    keep = date > 9204
    if shape == "array_bigint_column_ne":
        date = rng.integers(9200, 9208, npb).astype(np.int64)      # the range form of <> on a BIGINT column
        terms = [(1, abi.CMP_NE, 9204)]
        keep = date != 9204
    if shape == "array_partitioned_le":
        terms = [(1, abi.CMP_LE, 9204)]
        keep = date <= 9204
    key_cols, key_types = [0], [abi.BIGINT]
    build_cols, probe_cols = [bk, pay], [pk, date, price]
    dep_cols = [1]
    if shape == "hash":
        # a string key forces the generic hash mode
        key_types = [abi.VARCHAR]
        build_cols = [[b"k%015d" % v for v in bk], pay]
        probe_cols = [[b"k%015d" % v for v in pk], date, price]
    if shape == "two_keys_nullable_filter_column":
        k2b = rng.integers(0, 3, nb).astype(np.int32)
        k2p = rng.integers(0, 3, npb).astype(np.int32)
        key_cols, key_types = [0, 3], [abi.BIGINT, abi.INTEGER]
        build_cols = [bk, pay, pay, k2b]
        probe_cols = [pk, date, price, k2p]
        valid_date = rng.random(npb) > 0.1
        terms = [(1, abi.CMP_GT, 9204), (2, abi.CMP_LE, 100.0)]
        keep = (date > 9204) & valid_date & (price <= 100.0)     # a null fails its term
    sel = np.flatnonzero(keep)

    def host_batch(cols, rows=None):
        out = []
        for i, c in enumerate(cols):
            valid = valid_date if (i == 1 and valid_date is not None) else None
            if rows is not None:
                c = [c[j] for j in rows] if isinstance(c, list) else c[rows]
                valid = None if valid is None else valid[rows]
            out.append(c if valid is None else abi.HostColumn(abi.INTEGER, np.ascontiguousarray(c), np.ascontiguousarray(valid)))
        return batch_of(out)

    res = {}
    for impl in (oracle, vx):
        b = impl.JoinBuild(key_cols, key_types, dep_cols, [abi.BIGINT], join_type)
        b.add_input(batch_of(build_cols))
        table = b.finish()
        probe = impl.JoinProbe(table, key_cols, join_type)
        if impl is vx:
            probe.set_input_filter(terms)
            if shape == "dictionary_key":
                # the key column behind a dictionary (a pass-through column of an upstream operator)
                perm = rng.permutation(npb).astype(np.int32)
                inv = np.empty(npb, dtype=np.int32)
                inv[perm] = np.arange(npb, dtype=np.int32)
                hb = abi.HostBatch([abi.HostColumn(abi.BIGINT, pk[perm], None, abi.DICTIONARY, inv),
                                    abi.HostColumn(abi.INTEGER, date), abi.HostColumn(abi.DOUBLE, price)])
            else:
                hb = host_batch(probe_cols)
            vx.profile_reset()
            vx.profile_enable(True)
            probe.add_input(hb)
        else:
            probe.add_input(host_batch(probe_cols, sel))
        maps, rows, pays = [], [], []
        while True:
            m, r, cols, fin = probe.get_output(50001, [0])
            maps.append(np.asarray(m))
            rows.append(np.asarray(r))
            pays.append(np.where(np.asarray(cols[0][1]), np.asarray(cols[0][0]), -1))
            if fin:
                break
        mapping = np.concatenate(maps)
        if impl is oracle:
            mapping = sel[mapping]       # FilterProject's indices under the probe's
        out = [mapping, np.concatenate(rows), np.concatenate(pays)]
        if impl is vx:
            vx.profile_enable(False)
            names = set(vx.profile().keys())
            if shape.startswith("array_partitioned"):
                assert "k_join_probe_part" in names and "k_pp_scatter" in names
            if shape.startswith("array") or shape == "dictionary_key":
                assert table.stats().hash_mode == 1
            assert "k_filter_bits" not in names and "k_compact_write" not in names
        if join_type == abi.JOIN_RIGHT:
            br, bfin = [], False
            while not bfin:
                r2, c2, bfin = probe.get_build_side_output(60000, [0])
                br.append(np.asarray(r2))
            out.append(np.concatenate(br))
        res[impl.__name__] = out
    for g, e in zip(res[vx.__name__], res[oracle.__name__]):
        assert len(g) == len(e) and (g == e).all()
    assert len(res[vx.__name__][0]) > 1000


def test_input_filter_is_refused_where_unmatched_probe_rows_come_out(vx):
    t = vx.JoinBuild([0], [abi.BIGINT], [], [], abi.JOIN_LEFT)
    t.add_input(batch_of([np.arange(10, dtype=np.int64)]))
    table = t.finish()
    for jt in (abi.JOIN_LEFT,):
        p = vx.JoinProbe(table, [0], jt)
        with pytest.raises(vx.Vx355Error) as e:
            p.set_input_filter([(0, abi.CMP_GT, 3)])
        assert e.value.status == abi.EUNSUPPORTED
    for jt in (abi.JOIN_ANTI, abi.JOIN_FULL, abi.JOIN_LEFT_SEMI_PROJECT):
        b = vx.JoinBuild([0], [abi.BIGINT], [], [], jt)
        b.add_input(batch_of([np.arange(10, dtype=np.int64)]))
        p = vx.JoinProbe(b.finish(), [0], jt)
        with pytest.raises(vx.Vx355Error) as e:
            p.set_input_filter([(0, abi.CMP_GT, 3)])
        assert e.value.status == abi.EUNSUPPORTED
    # after the first batch it is too late
    b = vx.JoinBuild([0], [abi.BIGINT], [], [], abi.JOIN_INNER)
    b.add_input(batch_of([np.arange(10, dtype=np.int64)]))
    p = vx.JoinProbe(b.finish(), [0], abi.JOIN_INNER)
    p.add_input(batch_of([np.arange(5, dtype=np.int64)]))
    with pytest.raises(vx.Vx355Error):
        p.set_input_filter([(0, abi.CMP_GT, 3)])


def _regroup_case(rng, keys, nb, npr, miss_rate):
    """Build rows with sparse keys (normalized-key mode) and probe rows with misses inside and
    outside the build side's key range; keys = 1 (BIGINT) or 2 (BIGINT, INTEGER)."""
    bk = (rng.permutation(1 << 22)[:nb].astype(np.int64) * 104729 + 12345) * (1_000_003 if keys == 1 else 1)
    bk2 = rng.integers(0, 50, nb).astype(np.int32)    # (two keys: the product of the ranges must fit 2^59)
    pick = rng.integers(0, nb, npr)
    miss = rng.random(npr) < miss_rate
    pk = np.where(miss, np.where(rng.random(npr) < 0.5, bk[pick] + 1, -7), bk[pick]).astype(np.int64)
    pk2 = bk2[pick]
    if keys == 2:
        pk2 = np.where(rng.random(npr) < 0.1, 77, pk2).astype(np.int32)   # out of the second key's range
    return bk, bk2, pk, pk2


@pytest.mark.parametrize("join_type", [abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_ANTI, abi.JOIN_LEFT_SEMI_FILTER])
@pytest.mark.parametrize("keys,dups,wide_build,lds_build", [(1, False, "1", "1"), (1, False, "0", "0"), (1, True, "1", "1"),
                                                            (1, True, "0", "1"), (2, False, "0", "1"), (1, False, "1", "0")])
def test_probe_input_regrouped_by_table_slice(oracle, vx, join_type, keys, dups, wide_build, lds_build, monkeypatch):
    """vx355_join_probe_add_input_regrouped: every column of the batch (8-, 4-, 2-, 1- and 16-byte
    values) is moved so that rows probing the same slice of the slot array are adjacent, the operator
    probes THAT batch. Checked: the moved batch is a permutation of the input rows; mappings ascend and
    number the moved batch; the joined rows (probe row contents, build row) equal the oracle's as a
    multiset - unique and duplicate build keys, tables built wide (dependents inline) and narrow, one
    and two keys, misses inside and outside the key range, four join kinds, output taken in pages."""
    monkeypatch.setenv("VX355_JOIN_REGROUP", "1")
    monkeypatch.setenv("VX355_JOIN_SLICE_BYTES", "65536")    # 16-64 slices for a table of this size
    monkeypatch.setenv("VX355_JOIN_WIDE_BUILD", wide_build)
    monkeypatch.setenv("VX355_JOIN_LDS_BUILD", lds_build)     # the table assembled group by group in LDS, or slot by slot
    monkeypatch.setenv("VX355_JOIN_ARRAY_MAX", "0")
    rng = np.random.default_rng(1000 + keys + 2 * dups)
    nb, npr = 30_000, 150_001          # five histogram tiles, the last one partial; 74 probe units
    bk, bk2, pk, pk2 = _regroup_case(rng, keys, nb, npr, 0.2)
    if dups:
        bk[nb // 2:] = bk[: nb - nb // 2]
    d0 = rng.integers(-1 << 60, 1 << 60, nb).astype(np.int64)
    d1 = rng.random(nb)
    pv = rng.random(npr)
    pi = rng.integers(-1 << 30, 1 << 30, npr).astype(np.int32)
    ps = rng.integers(-1 << 14, 1 << 14, npr).astype(np.int16)
    pt = rng.integers(-100, 100, npr).astype(np.int8)
    pstr = [b"%09d" % i for i in range(npr)]        # inline strings: the row's own number
    key_cols = [0, 1][:keys]
    key_types = [abi.BIGINT, abi.INTEGER][:keys]
    build = batch_of([bk, bk2, d0, d1])
    probe_cols = [pk, pk2, pv, pi, ps, pt, pstr]
    probe_batch = batch_of(probe_cols)
    to, _ = _build(oracle, [[build]], key_cols, key_types, [2, 3], [abi.BIGINT, abi.DOUBLE], join_type)
    po = oracle.JoinProbe(to, key_cols, join_type)
    po.add_input(probe_batch)
    e_pairs, e_payload = _drain(po, 1 << 20)
    want = sorted((m, r, pay) for (m, r), pay in zip(e_pairs, e_payload))

    vx.profile_reset()
    vx.profile_enable(True)
    table, _b = _build(vx, [[vx.to_device(build)]], key_cols, key_types, [2, 3], [abi.BIGINT, abi.DOUBLE], join_type)
    vx.profile_enable(False)
    vx_build_prof = vx.profile()
    st = table.stats()
    assert st.hash_mode == abi.MODE_NORMALIZED_KEY and st.has_duplicates == (1 if dups else 0)
    assert st.num_distinct == len(np.unique(bk))
    probe = vx.JoinProbe(table, key_cols, join_type)
    widths = [8, 4, 8, 4, 2, 1, 16]
    outs = [vx.DeviceArray(npr * w, np.uint8) for w in widths]
    vx.profile_reset()
    vx.profile_enable(True)
    assert probe.add_input_regrouped(vx.to_device(probe_batch), [o.ptr for o in outs]) is True
    vx.profile_enable(False)
    prof = vx.profile()
    assert "k_grp_scatter" in prof and "k_join_probe_grouped" in prof and "k_widen_slots" not in prof
    assert ("k_lds_build" in vx_build_prof) == (lds_build == "1")
    moved = [o.to_host(npr * w) for o, w in zip(outs, widths)]
    mk = moved[0].view(np.int64)
    row_of = np.array([int(bytes(moved[6][i * 16 + 4: i * 16 + 13])) for i in range(npr)])   # StringView: size, 12 bytes
    assert sorted(row_of.tolist()) == list(range(npr))                  # a permutation of the input rows ...
    assert (mk == pk[row_of]).all() and (moved[1].view(np.int32) == pk2[row_of]).all()
    assert (moved[2].view(np.float64) == pv[row_of]).all() and (moved[3].view(np.int32) == pi[row_of]).all()
    assert (moved[4].view(np.int16) == ps[row_of]).all() and (moved[5].view(np.int8) == pt[row_of]).all()
    pairs, payload = _drain(probe, 40_000)                              # ... ascending mappings (asserted in _drain)
    _contiguous(pairs)
    got = sorted((int(row_of[m]), r, pay) for (m, r), pay in zip(pairs, payload))
    assert got == want      # (all rows of a duplicate chain are listed; sorted, so their order does not matter)
    assert len(got) > 10_000


def test_probe_input_is_not_regrouped_when_the_table_does_not_qualify(oracle, vx, monkeypatch):
    """Array-mode tables, small tables / batches under the adaptive rule, columns with null bitmaps:
    the batch is probed as it came, the caller's buffers stay untouched, results as add_input's."""
    rng = np.random.default_rng(5)
    nb, npr = 20_000, 60_000
    bk = rng.permutation(1 << 18)[:nb].astype(np.int64)
    pay = rng.integers(0, 1 << 40, nb).astype(np.int64)
    pk = rng.integers(0, 1 << 18, npr).astype(np.int64)
    pvalid = rng.random(npr) > 0.1
    for env, cols, valids in (("1", [pk], None), ("-1", [pk], None), ("1", [pk], [pvalid])):
        monkeypatch.setenv("VX355_JOIN_REGROUP", env)
        if valids is not None:
            monkeypatch.setenv("VX355_JOIN_ARRAY_MAX", "0")
        table, _ = _build(vx, [[batch_of([bk, pay])]], [0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER)
        a, b = vx.JoinProbe(table, [0], abi.JOIN_INNER), vx.JoinProbe(table, [0], abi.JOIN_INNER)
        out = vx.DeviceArray(np.full(npr, -1, dtype=np.int64))
        batch = vx.to_device(batch_of(cols, valids))
        assert a.add_input_regrouped(batch, [out.ptr]) is False
        b.add_input(batch)
        assert _drain(a, 50_000) == _drain(b, 50_000)
        assert (out.to_host(npr) == -1).all()


@pytest.mark.parametrize("wrapped", [False, True])
@pytest.mark.parametrize("shape", ["bigint_4_8", "two_keys_four_deps", "no_deps", "nullable_dep_falls_back",
                                   "five_deps_fall_back"])
def test_build_append_of_flat_and_dictionary_wrapped_integer_batches(oracle, vx, wrapped, shape):
    """HashBuild's append for the common build side (INTEGER / BIGINT keys, 4- or 8-byte dependents, FLAT or
    DICTIONARY, no nulls: k_build_append_flat, four rows per lane) against the oracle - batches whose row counts
    are not multiples of the kernel's 1024-row tile, several batches per build, both widths of keys and
    dependents, and shapes that must keep the interpreting kernel (a nullable dependent, more dependents than
    the specialised kernel takes). Semantics: HashBuild::addInput (exec/HashBuild.cpp:430-520)."""
    rng = np.random.default_rng(sum(shape.encode()) * 2 + int(wrapped))
    sizes = [1, 1023, 4097, 30000]
    two_keys = shape == "two_keys_four_deps"
    num_deps = {"bigint_4_8": 2, "two_keys_four_deps": 4, "no_deps": 0, "nullable_dep_falls_back": 2,
                "five_deps_fall_back": 5}[shape]
    dep_types = [abi.INTEGER if d % 2 == 0 else (abi.DOUBLE if d == 3 else abi.BIGINT) for d in range(num_deps)]
    key_types = [abi.BIGINT, abi.INTEGER] if two_keys else [abi.BIGINT]

    def column(kind, n, base_rows):
        if kind == abi.INTEGER:
            base = rng.integers(-50, 50, base_rows).astype(np.int32)
        elif kind == abi.DOUBLE:
            base = rng.random(base_rows)
        else:
            base = rng.integers(0, 5000, base_rows).astype(np.int64)
        return base

    batches = []
    for n in sizes:
        base_rows = 2 * n + 3 if wrapped else n
        idx = rng.integers(0, base_rows, n).astype(np.int32)
        cols = []
        for c, kind in enumerate(key_types + dep_types):
            base = column(kind, n, base_rows)
            valid = None
            if shape == "nullable_dep_falls_back" and c == len(key_types):
                valid = rng.random(n) > 0.2
            if wrapped:
                cols.append(abi.HostColumn(kind, base, valid, abi.DICTIONARY, idx))
            else:
                cols.append(abi.HostColumn(kind, base, valid))
        batches.append(abi.HostBatch(cols))
    nk = len(key_types)
    pk = [rng.integers(-100, 5100, 20000).astype(np.int64)]
    if two_keys:
        pk.append(rng.integers(-60, 60, 20000).astype(np.int32))
    res = {}
    for impl in (oracle, vx):
        table, _b = _build(impl, [batches], list(range(nk)), key_types, list(range(nk, nk + num_deps)), dep_types,
                           abi.JOIN_INNER)
        probe = impl.JoinProbe(table, list(range(nk)), abi.JOIN_INNER)
        probe.add_input(batch_of(pk))
        pairs, payload = _drain(probe, 4096)
        res[impl.__name__] = (_canon(pairs, payload), table.stats().num_distinct, table.stats().has_duplicates)
    assert res[oracle.__name__] == res[vx.__name__]
    assert len(res[vx.__name__][0]) > 0
