"""Helpers shared by the -m gpu parity tests."""
import math

import numpy as np

from velox_amd import abi


def batch_of(cols, valids=None):
    """numpy arrays -> HostBatch (kind from dtype; list of bytes -> VARCHAR;
    bool -> BOOLEAN)."""
    hc = []
    for i, c in enumerate(cols):
        valid = None if valids is None else valids[i]
        if isinstance(c, abi.HostColumn):
            hc.append(c)
            continue
        if isinstance(c, list):
            hc.append(abi.HostColumn(abi.VARCHAR, c, valid))
            continue
        c = np.asarray(c)
        kind = {np.dtype(np.int64): abi.BIGINT, np.dtype(np.int32): abi.INTEGER,
                np.dtype(np.float64): abi.DOUBLE, np.dtype(np.float32): abi.REAL,
                np.dtype(np.int16): abi.SMALLINT, np.dtype(np.int8): abi.TINYINT,
                np.dtype(bool): abi.BOOLEAN}[c.dtype]
        hc.append(abi.HostColumn(kind, c, valid))
    return abi.HostBatch(hc)


def run_agg(impl, batches, key_cols, key_types, aggs, step=abi.STEP_SINGLE, max_rows=777, **kw):
    op = impl.Aggregation(key_cols, key_types, aggs, step, **kw)
    for b in batches:
        op.add_input(b)
    op.no_more_input()
    return impl.collect_output(op, max_rows), op


def ulp_distance(a, b):
    """Distance in units in the last place between float64 arrays (NaN == NaN)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    ia = a.view(np.int64).copy()
    ib = b.view(np.int64).copy()
    ia = np.where(ia < 0, np.iinfo(np.int64).min - ia, ia)
    ib = np.where(ib < 0, np.iinfo(np.int64).min - ib, ib)
    d = np.abs(ia.astype(object) - ib.astype(object)).astype(np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.where(both_nan, 0, d)


def assert_columns_equal(got, exp, kinds, float_ulps=0, what=""):
    """Positional comparison of collect_output() results. Integer, key and
    validity columns must be identical; REAL/DOUBLE columns within float_ulps
    (0 = bit-identical, -0.0 == 0.0 allowed only when float_ulps > 0)."""
    assert len(got) == len(exp) == len(kinds), what
    for c, kind in enumerate(kinds):
        gv, gvalid = got[c]
        ev, evalid = exp[c]
        assert len(gvalid) == len(evalid), f"{what} col {c}: {len(gvalid)} vs {len(evalid)} rows"
        assert (np.asarray(gvalid) == np.asarray(evalid)).all(), f"{what} col {c}: validity"
        valid = np.asarray(evalid, dtype=bool)
        if kind in (abi.VARCHAR, abi.VARBINARY):
            assert [g for g, v in zip(gv, valid) if v] == [e for e, v in zip(ev, valid) if v], \
                f"{what} col {c}"
        elif kind in (abi.REAL, abi.DOUBLE):
            g = np.asarray(gv, dtype=np.float64)[valid]
            e = np.asarray(ev, dtype=np.float64)[valid]
            if float_ulps == 0:
                same = (g == e) | (np.isnan(g) & np.isnan(e))
                assert same.all(), f"{what} col {c}: {g[~same][:5]} vs {e[~same][:5]}"
            else:
                if kind == abi.REAL:
                    d = np.abs(g - e) / np.maximum(np.spacing(np.abs(e).astype(np.float32)), 1e-45)
                else:
                    d = ulp_distance(g, e)
                assert (d <= float_ulps).all(), f"{what} col {c}: max {d.max()} ulps"
        else:
            g = np.asarray(gv)[valid]
            e = np.asarray(ev)[valid]
            assert (g == e).all(), f"{what} col {c}"


def exact_group_sums(keys_tuple_list, values, valid=None):
    """math.fsum per group: the correctly rounded sum."""
    groups = {}
    for i, k in enumerate(keys_tuple_list):
        if valid is not None and not valid[i]:
            continue
        groups.setdefault(k, []).append(float(values[i]))
    return {k: math.fsum(v) for k, v in groups.items()}
