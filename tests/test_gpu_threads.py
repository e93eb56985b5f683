"""Operators of different Drivers run concurrently (exec/Driver.cpp:538: one thread per
Driver at a time, N Drivers in parallel; SURVEY.md §8(b) "Threading"). libvx355 has no
library-wide lock: every handle owns its HIP stream and mailbox, the HBM block cache is
shared. Four threads x four handles each, all results against the CPU oracle; ctypes
releases the GIL for the duration of every call, so the entry points really overlap."""
import threading

import numpy as np
import pytest

from velox_amd import abi
from gpu_util import assert_columns_equal, batch_of, run_agg

pytestmark = pytest.mark.gpu

AGGS = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT), (abi.AGG_MIN, 2, abi.BIGINT),
        (abi.AGG_SUM, 2, abi.BIGINT)]


def _agg_inputs(seed, groups):
    rng = np.random.default_rng(seed)
    n = 120000
    k = rng.integers(0, groups, n).astype(np.int64)
    v = rng.integers(0, 1 << 20, n).astype(np.float64) / 256
    w = rng.integers(-1 << 40, 1 << 40, n).astype(np.int64)
    return [batch_of([k[i:i + 30000], v[i:i + 30000], w[i:i + 30000]]) for i in range(0, n, 30000)]


def _join_inputs(seed):
    rng = np.random.default_rng(seed)
    bk = rng.permutation(200000)[:60000].astype(np.int64)
    bp = rng.integers(0, 1 << 30, 60000).astype(np.int64)
    pk = rng.integers(0, 200000, 150000).astype(np.int64)
    return bk, bp, pk


def _join(impl, bk, bp, pk):
    b = impl.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER)
    b.add_input(batch_of([bk, bp]))
    t = b.finish()
    p = impl.JoinProbe(t, [0], abi.JOIN_INNER)
    p.add_input(batch_of([pk]))
    maps, rows, pay = [], [], []
    while True:
        m, r, cols, fin = p.get_output(40000, [0])
        maps.append(np.asarray(m))
        rows.append(np.asarray(r))
        pay.append(np.asarray(cols[0][0]))
        if fin:
            break
    return np.concatenate(maps), np.concatenate(rows), np.concatenate(pay)


def test_four_threads_four_handles_each(oracle, vx):
    work = []
    for t in range(4):
        work.append([("agg", _agg_inputs(10 * t + 1, 500)), ("agg", _agg_inputs(10 * t + 2, 40000)),
                     ("join", _join_inputs(10 * t + 3)), ("agg", _agg_inputs(10 * t + 4, 7))])
    expected = []
    for items in work:
        exp = []
        for kind, data in items:
            if kind == "agg":
                exp.append(run_agg(oracle, data, [0], [abi.BIGINT], AGGS, max_rows=50000)[0])
            else:
                exp.append(_join(oracle, *data))
        expected.append(exp)
    results = [None] * 4
    errors = []
    start = threading.Barrier(4)

    def driver(t):
        try:
            start.wait()
            out = []
            for _ in range(3):   # three rounds: handles are created and destroyed while others run
                out = []
                # all four handles of this thread are alive at once, fed round-robin
                aggs = [(i, vx.Aggregation([0], [abi.BIGINT], AGGS)) for i, (kind, _) in enumerate(work[t])
                        if kind == "agg"]
                for step in range(4):
                    for i, op in aggs:
                        op.add_input(work[t][i][1][step])
                res = {}
                for i, op in aggs:
                    op.no_more_input()
                    res[i] = vx.collect_output(op, 50000)
                for i, (kind, data) in enumerate(work[t]):
                    out.append(res[i] if kind == "agg" else _join(vx, *data))
            results[t] = out
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=driver, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    kinds = [abi.BIGINT, abi.DOUBLE, abi.BIGINT, abi.BIGINT, abi.BIGINT]
    for t in range(4):
        for i, (kind, _) in enumerate(work[t]):
            if kind == "agg":
                assert_columns_equal(results[t][i], expected[t][i], kinds, what="thread %d op %d" % (t, i))
            else:
                for g, e in zip(results[t][i], expected[t][i]):
                    assert (g == e).all()


def test_one_join_table_probed_from_four_threads(oracle, vx):
    """N HashProbe instances share one const table (HashJoinBridge)."""
    bk, bp, _ = _join_inputs(77)
    b = vx.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER)
    b.add_input(batch_of([bk, bp]))
    table = b.finish()
    ob = oracle.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER)
    ob.add_input(batch_of([bk, bp]))
    otable = ob.finish()
    probes = [np.random.default_rng(500 + t).integers(0, 200000, 100000).astype(np.int64) for t in range(4)]

    def run(impl, tbl, pk):
        p = impl.JoinProbe(tbl, [0], abi.JOIN_INNER)
        outs = []
        for lo in range(0, len(pk), 25000):
            p.add_input(batch_of([pk[lo:lo + 25000]]))
            while True:
                m, r, cols, fin = p.get_output(9000, [0])
                outs.append((np.asarray(m) + lo, np.asarray(r), np.asarray(cols[0][0])))
                if fin:
                    break
        return [np.concatenate(x) for x in zip(*outs)]
    expected = [run(oracle, otable, pk) for pk in probes]
    results, errors = [None] * 4, []

    def driver(t):
        try:
            results[t] = run(vx, table, probes[t])
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))
    threads = [threading.Thread(target=driver, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(4):
        for g, e in zip(results[t], expected[t]):
            assert (g == e).all()
