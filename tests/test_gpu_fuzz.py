"""Differential fuzzing: random plans (key types and counts, encodings, nulls, masks, aggregate
sets, steps, batch sizes; join types, duplicates, payload types) through the MI355X library and
through the CPU oracle, same seeds, results compared exactly. Values are dyadic rationals so
that DOUBLE sums are exact and the comparison can be bit for bit."""
import numpy as np
import pytest

from velox_amd import abi
from gpu_util import assert_columns_equal, batch_of, run_agg

pytestmark = pytest.mark.gpu

WORDS = [b"", b"A", b"NO", b"RET", b"FURN", b"7 bytes", b"BUILDING", b"AUTOMOBILE", b"twelve bytes"]


def _values(rng, kind, n, card):
    if kind == abi.BOOLEAN:
        return rng.random(n) > 0.5
    if kind == abi.TINYINT:
        return rng.integers(-100, 100, n).astype(np.int8)
    if kind == abi.SMALLINT:
        return rng.integers(-3000, 3000, n).astype(np.int16)
    if kind == abi.INTEGER:
        return (rng.integers(0, card, n) * 7 - 1000).astype(np.int32)
    if kind == abi.BIGINT:
        # |values| < 2^31: avg(BIGINT) accumulates in double, 20 000 of them stay exact (< 2^53)
        pool = rng.integers(-2 ** 31, 2 ** 31, card) if rng.random() < 0.5 else np.arange(card) - 50
        return pool[rng.integers(0, card, n)].astype(np.int64)
    if kind == abi.REAL:
        return (rng.integers(-4096, 4096, n) / 16.0).astype(np.float32)
    if kind == abi.DOUBLE:
        return rng.integers(-1 << 20, 1 << 20, n) / 1024.0
    if kind == abi.TIMESTAMP:
        return np.stack([rng.integers(-5, 5, n), rng.integers(0, 3, n) * 1000], axis=1)
    if kind == abi.VARCHAR:
        top = int(rng.integers(3, len(WORDS) + 1))
        return [WORDS[i] for i in rng.integers(0, top, n)]
    raise AssertionError(kind)


def _skew(rng, values, n):
    """Sometimes most rows carry one value (hot keys are where LDS atomics contend)."""
    if n > 1 and rng.random() < 0.3:
        hot = rng.random(n) < rng.choice([0.6, 0.95])
        if isinstance(values, list):
            return [values[0] if h else v for h, v in zip(hot, values)]
        values = values.copy()
        values[hot] = values[0]
    return values


def _spec(rng, kind, n, card=200):
    """Raw material of a column of n rows: random encoding, random share of nulls."""
    enc = rng.choice(["flat", "flat", "dict", "const"]) if n > 0 else "flat"
    valid = None
    if rng.random() < 0.6:
        valid = rng.random(n) > rng.choice([0.02, 0.3])
    if enc == "dict":
        base = max(1, int(rng.integers(1, 50)))
        return dict(kind=kind, enc=enc, values=_values(rng, kind, base, card), indices=rng.integers(0, base, n),
                    valid=valid)
    if enc == "const":
        return dict(kind=kind, enc=enc, values=_values(rng, kind, 1, card), indices=None,
                    valid=None if valid is None else np.full(n, bool(valid[0])))
    return dict(kind=kind, enc=enc, values=_skew(rng, _values(rng, kind, n, card), n), indices=None, valid=valid)


def _build(spec, sel=None):
    """HostColumn of the spec, or of its rows 'sel' (the reference's FilterProject output: the
    same base vector wrapped in the surviving row numbers)."""
    kind, enc, values, indices, valid = spec["kind"], spec["enc"], spec["values"], spec["indices"], spec["valid"]
    if sel is not None:
        valid = None if valid is None else valid[sel]
        if enc == "flat":
            enc, indices = "dict", sel
        elif enc == "dict":
            indices = indices[sel]
    if enc == "dict":
        return abi.HostColumn(kind, values, valid, abi.DICTIONARY, indices)
    if enc == "const":
        return abi.HostColumn(kind, values, None if valid is None else valid[:1], abi.CONSTANT)
    return abi.HostColumn(kind, values, valid)


def _column(rng, kind, n, card=200):
    return _build(_spec(rng, kind, n, card))


@pytest.mark.parametrize("seed", range(400))
def test_random_aggregation_plans(oracle, vx, seed, monkeypatch):
    rng = np.random.default_rng(1000 + seed)
    if seed % 16 != 0:
        monkeypatch.setenv("VX355_JIT", "0")   # a hiprtc instantiation costs ~0.8 s per new plan shape
    if rng.random() < 0.3:
        monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    if rng.random() < 0.2:
        monkeypatch.setenv("VX355_AGG_RADIX_MIN_ROWS", "1")
    if rng.random() < 0.2:
        monkeypatch.setenv("VX355_ARRAY_MAX", str(int(rng.choice([0, 64, 4096]))))
    if rng.random() < 0.25:
        monkeypatch.setenv("VX355_AGG_CHUNK_ROWS", str(int(rng.choice([64, 4096]))))
    if rng.random() < 0.25:
        monkeypatch.setenv("VX355_AGG_DEFER_CAP", str(int(rng.choice([4, 1000]))))
    key_pool = [abi.BIGINT, abi.INTEGER, abi.SMALLINT, abi.TINYINT, abi.BOOLEAN, abi.VARCHAR, abi.DOUBLE, abi.REAL,
                abi.TIMESTAMP]
    num_keys = int(rng.integers(0, 4))
    key_types = [int(rng.choice(key_pool)) for _ in range(num_keys)]
    val_types = [int(rng.choice([abi.DOUBLE, abi.BIGINT, abi.INTEGER, abi.REAL, abi.SMALLINT]))
                 for _ in range(int(rng.integers(1, 4)))]
    has_mask = rng.random() < 0.3
    layout = key_types + val_types + ([abi.BOOLEAN] if has_mask else [])
    aggs = []
    for _ in range(int(rng.integers(0 if num_keys else 1, 6))):
        v = int(rng.integers(0, len(val_types)))
        fn = int(rng.choice([abi.AGG_SUM, abi.AGG_COUNT, abi.AGG_COUNT_STAR, abi.AGG_MIN, abi.AGG_MAX, abi.AGG_AVG]))
        col = -1 if fn == abi.AGG_COUNT_STAR else num_keys + v
        typ = abi.BIGINT if fn == abi.AGG_COUNT_STAR else val_types[v]
        mask = len(layout) - 1 if (has_mask and rng.random() < 0.5) else -1
        aggs.append((fn, col, typ, mask))
    card = int(rng.choice([3, 40, 3000]))
    # A fused FilterProject in front (vx355_agg_set_fused_input): filter on an extra INTEGER column;
    # the oracle sees what the reference's FilterProject would hand over (the surviving rows).
    fused = rng.random() < 0.3
    cut = int(rng.integers(10, 90))
    batches, filtered = [], []
    for _ in range(int(rng.integers(1, 5))):
        n = int(rng.choice([1, 63, 64, 1000, 5000, 20000, 150000]))
        specs = [_spec(rng, k, n, card) for k in layout]
        cols = [_build(sp) for sp in specs]
        if fused:
            f = rng.integers(0, 100, n).astype(np.int32)
            sel = np.flatnonzero(f <= cut).astype(np.int32)
            batches.append(abi.HostBatch(cols + [abi.HostColumn(abi.INTEGER, f)], n))
            if len(sel):
                filtered.append(abi.HostBatch([_build(sp, sel) for sp in specs], len(sel)))
        else:
            batches.append(abi.HostBatch(cols, n))
    kw = dict(ignore_null_keys=bool(rng.random() < 0.3))
    key_cols = list(range(num_keys))
    if fused:
        exp, _ = run_agg(oracle, filtered, key_cols, key_types, aggs, max_rows=100000, **kw)
        op = vx.Aggregation(key_cols, key_types, aggs, abi.STEP_SINGLE, **kw)
        op.set_fused_input([(len(layout), abi.CMP_LE, cut)], [])
        for bt in batches:
            op.add_input(bt)
        op.no_more_input()
        got = vx.collect_output(op, int(rng.choice([7, 1000, 100000])))
        assert_columns_equal(got, exp, op.kinds, what=f"seed {seed}: fused, keys {key_types} aggs {aggs}")
        return
    exp, _ = run_agg(oracle, batches, key_cols, key_types, aggs, max_rows=100000, **kw)
    got, gop = run_agg(vx, batches, key_cols, key_types, aggs, max_rows=int(rng.choice([7, 1000, 100000])), **kw)
    assert_columns_equal(got, exp, gop.kinds, what=f"seed {seed}: keys {key_types} aggs {aggs}")
    if rng.random() < 0.5 and aggs:
        # partial per batch -> final over the partial outputs == single (docs/develop/aggregations.rst)
        from velox_amd import dist as vdist
        raw = [(a[0], a[1], a[2], a[3]) for a in aggs]
        parts = []
        for b in batches:
            out, op = run_agg(vx, [b], key_cols, key_types, raw, step=abi.STEP_PARTIAL, max_rows=100000, **kw)
            parts.append((out, op.kinds))
        def as_batch(outputs):
            """collect_output() results of the same layout -> one HostBatch (rows concatenated)."""
            cols = []
            for c, kind in enumerate(outputs[0][1]):
                vals = [p[0][c][0] for p in outputs]
                valid = np.concatenate([np.asarray(p[0][c][1], dtype=bool) for p in outputs])
                if kind == abi.VARCHAR:
                    vals = [v if v is not None else b"" for v in sum((list(v) for v in vals), [])]
                else:
                    vals = np.concatenate(vals)
                cols.append(abi.HostColumn(kind, vals, valid))
            return abi.HostBatch(cols, len(cols[0].valid))
        fin_aggs = vdist.final_aggs_for([(a[0], a[1], a[2]) for a in raw], num_keys)
        if rng.random() < 0.5 and len(parts) > 1:
            # partial -> intermediate (intermediate layout in and out) -> final
            mid = []
            for group in (parts[:len(parts) // 2], parts[len(parts) // 2:]):
                out, op = run_agg(vx, [as_batch(group)], key_cols, key_types, fin_aggs, step=abi.STEP_INTERMEDIATE,
                                  max_rows=100000, **kw)
                assert op.kinds == group[0][1]
                mid.append((out, op.kinds))
            parts = mid
        merged, mop = run_agg(vx, [as_batch(parts)], key_cols, key_types, fin_aggs, step=abi.STEP_FINAL,
                              max_rows=100000, **kw)
        assert_columns_equal(merged, exp, mop.kinds, what=f"seed {seed}: partial -> final")


def test_low_cardinality_plans_with_nulls_report_their_kernel(oracle, vx, monkeypatch, capsys):
    """Which kernel do low-cardinality plans over flat, nullable columns take? 32 random plans: one or two
    keys (INTEGER / BIGINT / strings of <= 3 bytes, nullable), two to four aggregates from sum / avg / min /
    max / count over DOUBLE / REAL / BIGINT / INTEGER operands (nullable), count(*), FILTER masks, sometimes
    a fused filter term; every result equals the oracle's, and the census of kernels is printed. At least
    90 % of them must run on the shape-specialised kernel (k_agg_fast), not on the interpreting one."""
    monkeypatch.setenv("VX355_JIT", "sync")
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    census = {}
    slow = []
    plans = 32
    for seed in range(plans):
        rng = np.random.default_rng(7000 + seed)
        n = 60_000
        num_keys = int(rng.integers(1, 3))
        key_types = [int(rng.choice([abi.INTEGER, abi.BIGINT, abi.VARCHAR])) for _ in range(num_keys)]
        val_types = [int(rng.choice([abi.DOUBLE, abi.REAL, abi.BIGINT, abi.INTEGER])) for _ in range(int(rng.integers(1, 4)))]
        cols, valids, kinds = [], [], []
        for kt in key_types:
            card = int(rng.choice([3, 12, 40]))
            if kt == abi.VARCHAR:
                words = [b"A", b"NO", b"RET", b"", b"F", b"O"][:max(2, card % 7)]
                cols.append([words[i] for i in rng.integers(0, len(words), n)])
            else:
                cols.append((rng.integers(0, card, n) * 3 - 7).astype(np.int32 if kt == abi.INTEGER else np.int64))
            valids.append(rng.random(n) > 0.03 if rng.random() < 0.5 else None)
            kinds.append(kt)
        for vt in val_types:
            cols.append(_values(rng, vt, n, 200))
            valids.append(rng.random(n) > float(rng.choice([0.01, 0.2])) if rng.random() < 0.7 else None)
            kinds.append(vt)
        mask_cols = []
        for _ in range(int(rng.integers(0, 3))):
            mask_cols.append(len(cols))
            cols.append(rng.random(n) > 0.5)
            valids.append(rng.random(n) > 0.1 if rng.random() < 0.5 else None)
            kinds.append(abi.BOOLEAN)
        aggs = []
        for _ in range(int(rng.integers(2, 5))):
            v = int(rng.integers(0, len(val_types)))
            fn = int(rng.choice([abi.AGG_SUM, abi.AGG_AVG, abi.AGG_MIN, abi.AGG_MAX, abi.AGG_COUNT, abi.AGG_COUNT_STAR]))
            col = -1 if fn == abi.AGG_COUNT_STAR else num_keys + v
            typ = abi.BIGINT if fn == abi.AGG_COUNT_STAR else val_types[v]
            mask = int(rng.choice(mask_cols)) if (mask_cols and rng.random() < 0.4) else -1
            aggs.append((fn, col, typ, mask))
        fused = rng.random() < 0.4
        kw = dict(ignore_null_keys=bool(rng.random() < 0.3))
        key_cols = list(range(num_keys))
        if fused:
            f = rng.integers(0, 100, n).astype(np.int32)
            fvalid = rng.random(n) > 0.02
            cut = int(rng.integers(10, 90))
            keep = np.flatnonzero((f <= cut) & fvalid)

            def take(c):
                return [c[i] for i in keep] if isinstance(c, list) else np.asarray(c)[keep]
            ref = batch_of([take(c) for c in cols], [None if v is None else v[keep] for v in valids])
            exp, _ = run_agg(oracle, [ref], key_cols, key_types, aggs, max_rows=100000, **kw)
            host = batch_of(cols + [f], valids + [fvalid])
        else:
            host = batch_of(cols, valids)
            exp, _ = run_agg(oracle, [host], key_cols, key_types, aggs, max_rows=100000, **kw)
        op = vx.Aggregation(key_cols, key_types, aggs, abi.STEP_SINGLE, **kw)
        if fused:
            op.set_fused_input([(len(cols), abi.CMP_LE, cut)], [])
        vx.profile_reset()
        vx.profile_enable(True)
        op.add_input(vx.to_device(host))
        op.no_more_input()
        got = vx.collect_output(op, 100000)
        vx.profile_enable(False)
        assert_columns_equal(got, exp, op.kinds, what=f"census plan {seed}: keys {key_types} aggs {aggs}")
        names = vx.profile()
        took = "k_agg_fast" if ("k_agg_fast" in names and "k_agg_lds" not in names) else \
            ("k_agg_lds" if "k_agg_lds" in names else "+".join(sorted(k for k in names if k.startswith("k_agg"))))
        census[took] = census.get(took, 0) + 1
        if took != "k_agg_fast":
            slow.append((seed, key_types, val_types, aggs, fused))
    with capsys.disabled():
        print("\nkernel census of %d low-cardinality plans with nulls / masks: %s" % (plans, census))
        for item in slow:
            print("  not on the specialised kernel: seed %d keys %s operands %s aggs %s fused %s" % item)
    assert census.get("k_agg_fast", 0) >= 0.9 * plans, census


LONG_WORDS = WORDS + [b"thirteen bytes", b"a string well beyond the inline limit of a view", b"a string well beyond the inline limit",
                      b"twelve bytes\x00", b"\xff\xff", b"A\x00"]


@pytest.mark.parametrize("seed", range(200))
def test_random_distinct_and_string_aggregates(oracle, vx, seed, monkeypatch):
    """DISTINCT sum / count / avg and min / max over VARCHAR mixed with plain aggregates: random
    key sets (every table mode), encodings, nulls, masks, batch counts; SINGLE step, exact compare."""
    rng = np.random.default_rng(77000 + seed)
    monkeypatch.setenv("VX355_JIT", "0")
    if rng.random() < 0.3:
        monkeypatch.setenv("VX355_ARRAY_MAX", str(int(rng.choice([0, 64, 4096]))))
    if rng.random() < 0.3:
        monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    key_pool = [abi.BIGINT, abi.INTEGER, abi.SMALLINT, abi.BOOLEAN, abi.VARCHAR, abi.DOUBLE]
    num_keys = int(rng.integers(0, 4))
    key_types = [int(rng.choice(key_pool)) for _ in range(num_keys)]
    val_types = [int(rng.choice([abi.DOUBLE, abi.BIGINT, abi.INTEGER, abi.REAL])) for _ in range(2)] + [abi.VARCHAR]
    has_mask = rng.random() < 0.5
    layout = key_types + val_types + ([abi.BOOLEAN] if has_mask else [])
    S = num_keys + 2
    aggs = []
    for _ in range(int(rng.integers(1, 6))):
        mask = len(layout) - 1 if (has_mask and rng.random() < 0.5) else -1
        what = rng.random()
        if what < 0.35:
            aggs.append((int(rng.choice([abi.AGG_MIN, abi.AGG_MAX])), S, abi.VARCHAR, mask))
        else:
            v = int(rng.integers(0, 2))
            fn = int(rng.choice([abi.AGG_SUM, abi.AGG_COUNT, abi.AGG_AVG, abi.AGG_MIN, abi.AGG_MAX]))
            flags = abi.AGG_FN_DISTINCT if rng.random() < 0.7 else 0
            aggs.append((fn, num_keys + v, val_types[v], mask, -1, flags))
    card = int(rng.choice([3, 40, 3000]))
    batches = []
    for _ in range(int(rng.integers(1, 4))):
        n = int(rng.choice([1, 64, 1000, 5000, 30000]))
        cols = []
        for j, k in enumerate(layout):
            sp = _spec(rng, k, n, card)
            if j == S:  # strings on both sides of the inline limit for min / max
                m = len(sp["values"]) if isinstance(sp["values"], list) else n
                sp["values"] = [LONG_WORDS[i] for i in rng.integers(0, len(LONG_WORDS), m)]
            cols.append(_build(sp))
        batches.append(abi.HostBatch(cols, n))
    kw = dict(ignore_null_keys=bool(rng.random() < 0.3))
    key_cols = list(range(num_keys))
    # DISTINCT exists in the SINGLE step only; the other plans also run as PARTIAL (avg = sum, count)
    step = abi.STEP_SINGLE
    if not any(len(a) > 5 and a[5] for a in aggs) and rng.random() < 0.4:
        step = abi.STEP_PARTIAL
    exp, eop = run_agg(oracle, batches, key_cols, key_types, aggs, step, max_rows=100000, **kw)
    got, gop = run_agg(vx, batches, key_cols, key_types, aggs, step, max_rows=int(rng.choice([7, 1000, 100000])), **kw)
    assert gop.kinds == eop.kinds
    assert_columns_equal(got, exp, gop.kinds, what=f"seed {seed}: step {step} keys {key_types} aggs {aggs}")


def _canon_join(probe, join_type, max_rows):
    rows = []
    while True:
        mapping, build_rows, cols, fin = probe.get_output(max_rows)
        for i in range(len(mapping)):
            pay = tuple(None if not valid[i] else (vals[i] if isinstance(vals, list) else vals[i].item())
                        for vals, valid in cols)
            if join_type == abi.JOIN_LEFT_SEMI_PROJECT:
                rows.append((int(mapping[i]), bool(build_rows[i] >= 0)))
            else:
                rows.append((int(mapping[i]), pay))
        if fin:
            break
    assert [r[0] for r in rows] == sorted(r[0] for r in rows)       # ascending probe rows
    out = [sorted(rows, key=repr)]
    if join_type in (abi.JOIN_RIGHT, abi.JOIN_FULL, abi.JOIN_RIGHT_SEMI_FILTER):
        side = []
        while True:
            build_rows, cols, fin = probe.get_build_side_output(max_rows)
            for i in range(len(build_rows)):
                side.append((int(build_rows[i]),) + tuple(
                    None if not valid[i] else (vals[i] if isinstance(vals, list) else vals[i].item())
                    for vals, valid in cols))
            if fin:
                break
        out.append(side)
    return out


@pytest.mark.parametrize("seed", range(60))
def test_random_join_plans(oracle, vx, seed, monkeypatch):
    rng = np.random.default_rng(5000 + seed)
    if rng.random() < 0.3:
        monkeypatch.setenv("VX355_JOIN_ARRAY_MAX", "0")
    join_type = int(rng.choice([abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_RIGHT, abi.JOIN_FULL, abi.JOIN_LEFT_SEMI_FILTER,
                                abi.JOIN_LEFT_SEMI_PROJECT, abi.JOIN_RIGHT_SEMI_FILTER, abi.JOIN_ANTI]))
    null_aware = join_type == abi.JOIN_ANTI and rng.random() < 0.5
    key_types = [int(rng.choice([abi.BIGINT, abi.INTEGER, abi.VARCHAR, abi.SMALLINT, abi.DOUBLE]))
                 for _ in range(int(rng.integers(1, 3)))]
    dep_types = [int(rng.choice([abi.BIGINT, abi.DOUBLE, abi.INTEGER, abi.VARCHAR, abi.BOOLEAN, abi.REAL]))
                 for _ in range(int(rng.integers(0, 4)))]
    card = int(rng.choice([5, 300, 5000]))
    nk = len(key_types)
    results = {}
    build_batches = []
    for _ in range(int(rng.integers(1, 3))):       # build drivers
        batches = []
        for _ in range(int(rng.integers(1, 3))):
            n = int(rng.choice([0, 1, 200, 3000])) if rng.random() < 0.9 else 0
            batches.append(abi.HostBatch([_column(rng, k, n, card) for k in key_types + dep_types], n))
        build_batches.append(batches)
    if null_aware and rng.random() < 0.5:
        # keep the build side free of null keys so that the regular NOT IN branch is exercised
        build_batches = [[abi.HostBatch([abi.HostColumn(k, _values(rng, k, 500, card)) for k in key_types + dep_types], 500)]]
    probes = []
    for _ in range(int(rng.integers(1, 3))):
        n = int(rng.choice([1, 64, 2000, 9000]))
        probes.append(abi.HostBatch([_column(rng, k, n, card) for k in key_types], n))
    for impl in (oracle, vx):
        builds = []
        for batches in build_batches:
            b = impl.JoinBuild(list(range(nk)), key_types, list(range(nk, nk + len(dep_types))), dep_types, join_type,
                               null_aware)
            for hb in batches:
                b.add_input(hb)
            builds.append(b)
        table = builds[0].finish(builds[1:])
        handles = [impl.JoinProbe(table, list(range(nk)), join_type, null_aware) for _ in probes]
        out = []
        for h, pb in zip(handles, probes):
            h.add_input(pb)
            part = _canon_join(h, join_type, int(rng.choice([50, 1000])) if impl is vx else 1000) \
                if h is handles[-1] else _canon_join_probe_only(h, join_type)
            out.append(part)
        results[impl.__name__] = out
    assert results[oracle.__name__] == results[vx.__name__], f"seed {seed}: {join_type} keys {key_types} deps {dep_types}"


def _canon_join_probe_only(probe, join_type):
    """Drains the probe-side output of a handle that is not the last prober."""
    rows = []
    while True:
        mapping, build_rows, cols, fin = probe.get_output(777)
        for i in range(len(mapping)):
            pay = tuple(None if not valid[i] else (vals[i] if isinstance(vals, list) else vals[i].item())
                        for vals, valid in cols)
            rows.append((int(mapping[i]), bool(build_rows[i] >= 0)) if join_type == abi.JOIN_LEFT_SEMI_PROJECT
                        else (int(mapping[i]), pay))
        if fin:
            break
    return [sorted(rows, key=repr)]
