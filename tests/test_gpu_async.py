"""The asynchronous boundary (include/vx355.h, ABI 5): *_add_input_async queues a batch for the handle's
worker thread and returns; poll / wait report progress and failures; every other entry point drains the
queue first. Results must equal the synchronous path (same batches, same order -> same first-seen group
order) and the oracle."""
import numpy as np
import pytest

from velox_amd import abi
from gpu_util import assert_columns_equal, batch_of, run_agg

pytestmark = pytest.mark.gpu


def _batches(rng, count, rows):
    out = []
    for _ in range(count):
        k = rng.integers(0, 500, rows).astype(np.int64)
        v = rng.integers(-1 << 20, 1 << 20, rows) / 1024.0
        w = rng.integers(-1000, 1000, rows).astype(np.int64)
        out.append(batch_of([k, v, w], [None, rng.random(rows) > 0.05, None]))
    return out


@pytest.mark.parametrize("device_resident", [False, True])
def test_aggregation_async_input_equals_the_synchronous_path(oracle, vx, device_resident):
    rng = np.random.default_rng(11)
    batches = _batches(rng, 120, 5000)
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_SUM, 2, abi.BIGINT), (abi.AGG_MIN, 2, abi.BIGINT),
            (abi.AGG_COUNT, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, batches, [0], [abi.BIGINT], aggs, max_rows=4096)
    feed = [vx.to_device(b) for b in batches] if device_resident else batches
    op = vx.Aggregation([0], [abi.BIGINT], aggs)
    tickets = [op.add_input_async(b) for b in feed[:100]]
    assert tickets == list(range(1, 101))
    submitted, completed = op.poll()
    assert submitted == 100 and 0 <= completed <= 100
    op.add_input(feed[100])            # a synchronous call drains the queue first: order is kept
    assert op.poll() == (100, 100)
    for b in feed[101:]:
        op.add_input_async(b)
    op.no_more_input()                 # ... and so does noMoreInput
    assert op.poll() == (119, 119)
    got = vx.collect_output(op, 4096)
    assert_columns_equal(got, exp, op.kinds, what="async input")
    assert op.stats().input_rows == 120 * 5000


def test_a_failed_batch_is_reported_by_wait_and_poisons_the_ones_behind_it(vx):
    rng = np.random.default_rng(12)
    good = _batches(rng, 3, 1000)
    bad = batch_of([rng.integers(0, 5, 10).astype(np.int64)])     # the plan reads columns 1 and 2 as well
    op = vx.Aggregation([0], [abi.BIGINT], [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_SUM, 2, abi.BIGINT)])
    op.add_input_async(good[0])
    op.add_input_async(bad)
    op.add_input_async(good[1])
    with pytest.raises(vx.Vx355Error) as e:
        op.wait()
    assert e.value.status == abi.EINVAL
    assert op.poll() == (3, 3)
    assert op.stats().input_rows == 1000   # the batch behind the failure was skipped (stats stay readable)
    # the failure stays with the handle: the operator is short of input, so nothing may succeed quietly
    for call in (op.wait, lambda: op.add_input_async(good[2]), lambda: op.add_input(good[2]), op.no_more_input,
                 lambda: vx.collect_output(op, 100)):
        with pytest.raises(vx.Vx355Error) as again:
            call()
        assert again.value.status == abi.EINVAL


def test_join_build_async_input(oracle, vx):
    rng = np.random.default_rng(13)
    nb = 40
    builds = []
    for i in range(nb):
        k = (np.arange(i * 2000, (i + 1) * 2000, dtype=np.int64) * 7) % 100_003
        builds.append(batch_of([k, rng.integers(0, 1 << 40, 2000).astype(np.int64)]))
    pk = rng.integers(0, 100_003, 50_000).astype(np.int64)

    def run(impl, asynchronous):
        b = impl.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER)
        for hb in builds:
            if asynchronous:
                b.add_input_async(hb)
            else:
                b.add_input(hb)
        t = b.finish()                  # finish waits for the queue
        p = impl.JoinProbe(t, [0], abi.JOIN_INNER)
        p.add_input(batch_of([pk]))
        rows = []
        while True:
            m, r, cols, fin = p.get_output(1 << 16)
            rows += list(zip(np.asarray(m).tolist(), np.asarray(cols[0][0]).tolist()))
            if fin:
                break
        return sorted(rows)
    want = run(oracle, False)
    assert run(vx, True) == want and len(want) > 0
