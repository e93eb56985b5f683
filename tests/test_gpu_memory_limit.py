"""What happens beyond the operators' share of HBM (include/vx355.h, "memory beyond the operator's
share"). The reference's operators reclaim / spill or fail the query when their MemoryPool is
exhausted (exec/HashBuild.cpp:995 ensureTableFits, :1314 reclaim; exec/HashAggregation.cpp:562;
exec/HashProbe.cpp:2182); this library does not spill, so the contract is the failing half: the entry
point that needed the memory returns VX355_ENOMEM, the handle stays destroyable, destroy gives
everything back, other operators keep working, and the same plan succeeds once the limit is lifted."""
import gc

import numpy as np
import pytest

from velox_amd import abi
from gpu_util import batch_of, run_agg

pytestmark = pytest.mark.gpu

MIB = 1 << 20


@pytest.fixture
def limited(vx):
    gc.collect()
    yield vx
    vx.set_memory_limit(0)


def _held(vx):
    gc.collect()
    return vx.memory_usage()[0]


def test_join_build_beyond_the_limit_fails_cleanly_and_releases_everything(oracle, limited):
    vx = limited
    rng = np.random.default_rng(1)
    n = 3_000_000
    keys = rng.permutation(4 * n)[:n].astype(np.int64) * 7919        # sparse: a normalized-key table of tens of MB
    pay = rng.integers(0, 1 << 40, n).astype(np.int64)
    probe_keys = keys[rng.integers(0, n, 200_000)]
    before = _held(vx)
    vx.set_memory_limit(before + 24 * MIB)
    build = vx.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER)
    with pytest.raises(vx.Vx355Error) as e:
        for lo in range(0, n, 500_000):
            build.add_input(batch_of([keys[lo:lo + 500_000], pay[lo:lo + 500_000]]))
        build.finish()
    assert e.value.status == abi.ENOMEM and "memory limit" in str(e.value)
    del e       # (the traceback keeps the frame of build.add_input, and with it the handle, alive)
    # a second operator, small enough, works while the failed one is still around ...
    small = vx.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER)
    small.add_input(batch_of([keys[:1000], pay[:1000]]))
    table = small.finish()
    probe = vx.JoinProbe(table, [0], abi.JOIN_INNER)
    probe.add_input(batch_of([keys[:50]]))
    mapping, rows, _, fin = probe.get_output(1000, [])
    assert fin and list(mapping) == list(range(50)) and list(rows) == list(range(50))
    del probe, table, small
    # ... the failed handle is destroyable and gives everything back
    del build
    assert _held(vx) == before
    # the same plan with the limit lifted: the oracle's answer
    vx.set_memory_limit(0)

    def run(impl):
        b = impl.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER)
        b.add_input(batch_of([keys, pay]))
        p = impl.JoinProbe(b.finish(), [0], abi.JOIN_INNER)
        p.add_input(batch_of([probe_keys]))
        out_rows, out_pay = [], []
        while True:
            mapping, rows, cols, fin = p.get_output(1 << 20, [0])
            out_rows.append(np.asarray(rows))
            out_pay.append(np.asarray(cols[0][0]))
            if fin:
                return np.concatenate(out_rows), np.concatenate(out_pay)
    got, want = run(vx), run(oracle)
    assert (got[0] == want[0]).all() and (got[1] == want[1]).all()
    assert _held(vx) == before


def test_aggregation_beyond_the_limit_fails_cleanly(oracle, limited):
    vx = limited
    rng = np.random.default_rng(2)
    n = 4_000_000
    keys = rng.permutation(8 * n)[:n].astype(np.int64) * 104729      # every row its own group
    vals = rng.integers(-1000, 1000, n).astype(np.int64)
    aggs = [(abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    before = _held(vx)
    vx.set_memory_limit(before + 32 * MIB)
    op = vx.Aggregation([0], [abi.BIGINT], aggs)
    with pytest.raises(vx.Vx355Error) as e:
        for lo in range(0, n, 500_000):
            op.add_input(batch_of([keys[lo:lo + 500_000], vals[lo:lo + 500_000]]))
        op.no_more_input()
        vx.collect_output(op, 1 << 20)
    assert e.value.status == abi.ENOMEM
    del e
    del op
    assert _held(vx) == before
    # under the same limit a plan that fits still runs, and matches the oracle
    few = keys[:200_000] % 1000
    got, _ = run_agg(vx, [batch_of([few, vals[:200_000]])], [0], [abi.BIGINT], aggs)
    want, _ = run_agg(oracle, [batch_of([few, vals[:200_000]])], [0], [abi.BIGINT], aggs)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert (np.asarray(g[0]) == np.asarray(w[0])).all()
    assert _held(vx) == before


def test_usage_counters_follow_the_operators(limited):
    vx = limited
    before = _held(vx)
    vx.memory_usage()                        # (resets the peak)
    rng = np.random.default_rng(3)
    keys = rng.permutation(2_000_000).astype(np.int64) * 31
    build = vx.JoinBuild([0], [abi.BIGINT], [], [], abi.JOIN_INNER)
    build.add_input(batch_of([keys]))
    table = build.finish()
    held, peak, _ = vx.memory_usage()
    assert held > before + 16 * MIB and peak >= held         # the table (>= 16 B a row) is counted while it lives
    del table, build
    assert _held(vx) == before
