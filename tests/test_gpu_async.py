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


@pytest.mark.parametrize("chunk_rows", [None, "40000"])
def test_parallel_ingest_of_small_host_vectors_keeps_order_and_results(oracle, vx, monkeypatch, chunk_rows):
    """Host vectors without null bitmaps go through the parallel ingest (copier threads fill pinned
    chunks, the worker takes whole chunks): same groups in the same first-seen order as the oracle fed
    the same vectors one by one. In the middle of the stream: a vector with a null bitmap (ordinary
    path, in order), one with a 20-byte string (its chunk falls back to vector-by-vector input) and
    an empty one. Small chunks (VX355_INGEST_CHUNK_ROWS) make the ring of four chunks go round."""
    if chunk_rows:
        monkeypatch.setenv("VX355_INGEST_CHUNK_ROWS", chunk_rows)
    rng = np.random.default_rng(21)
    codes = [b"AA", b"B", b"CCC", b"", b"DDDDDDDDDDDD"]
    batches = []
    for i in range(150):
        n = 0 if i == 70 else int(rng.integers(3000, 9000))
        # new keys keep appearing: the first-seen order depends on the order the vectors take effect
        k = rng.integers(0, 50 + 20 * i, n).astype(np.int64)
        s = [codes[j] for j in rng.integers(0, len(codes), n)]
        if i == 90 and n:
            s[5] = b"a string of 20 bytes"
        v = rng.integers(-1 << 20, 1 << 20, n) / 1024.0
        valid = [None, None, (rng.random(n) > 0.1) if i == 40 else None]
        batches.append(batch_of([k, s, v], valid))
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_COUNT, 2, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, batches, [0, 1], [abi.BIGINT, abi.VARCHAR], aggs, max_rows=1 << 20)
    op = vx.Aggregation([0, 1], [abi.BIGINT, abi.VARCHAR], aggs)
    tickets = [op.add_input_async(b) for b in batches]
    assert tickets == list(range(1, 151))
    sub, done = op.poll()
    assert sub == 150 and 0 <= done <= 150
    op.wait()
    assert op.poll() == (150, 150)
    op.no_more_input()
    got = vx.collect_output(op, 1 << 20)
    assert_columns_equal(got, exp, op.kinds, what="parallel ingest")
    assert op.stats().input_rows == sum(b.num_rows for b in batches)


def test_polling_alone_completes_the_batches_of_an_open_chunk(vx):
    """A shim that never calls wait: the batches assigned to a chunk that is not full yet complete
    once polls see no new submissions (the open chunk is handed to the worker)."""
    import time
    rng = np.random.default_rng(22)
    op = vx.Aggregation([0], [abi.BIGINT], [(abi.AGG_COUNT_STAR, -1, abi.BIGINT)])
    for _ in range(3):
        op.add_input_async(batch_of([rng.integers(0, 10, 1000).astype(np.int64)]))
    deadline = time.time() + 30
    while op.poll() != (3, 3):
        assert time.time() < deadline, op.poll()
        time.sleep(0.001)
    op.no_more_input()
    got = vx.collect_output(op, 100)
    assert int(np.sum(got[1][0])) == 3000


def test_probe_async_input_equals_the_synchronous_path(oracle, vx):
    """vx355_join_probe_add_input_async (ABI 7): the batch's upload and probe kernels run on the handle's
    worker thread; get_output waits for them. Batch after batch, mixed with the synchronous form, with an
    input filter fused in: the same (probe row, build row, payload) lists as the oracle's."""
    rng = np.random.default_rng(17)
    nb = 60000
    bk = rng.permutation(4_000_000)[:nb].astype(np.int64)
    pay = rng.integers(0, 1 << 30, nb).astype(np.int64)
    probes = []
    for _ in range(6):
        pk = rng.integers(0, 4_000_000, 300_000).astype(np.int64)
        pk[::20] = bk[rng.integers(0, nb, len(pk[::20]))]
        probes.append((pk, rng.integers(0, 100, len(pk)).astype(np.int32)))
    res = {}
    for impl in (oracle, vx):
        b = impl.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER)
        b.add_input(batch_of([bk, pay]))
        probe = impl.JoinProbe(b.finish(), [0], abi.JOIN_INNER)
        if impl is vx:
            probe.set_input_filter([(1, abi.CMP_LT, 50)])
        out = []
        for i, (pk, flag) in enumerate(probes):
            if impl is vx:
                hb = batch_of([pk, flag])
                if i % 3 == 2:
                    probe.add_input(hb)
                else:
                    ticket = probe.add_input_async(hb)
                    assert ticket >= 1
                    sub, done = probe.poll()
                    assert sub == ticket and done in (ticket - 1, ticket)
                    if i % 3 == 1:
                        probe.wait()
                        assert probe.poll() == (ticket, ticket)
                sel = None
            else:
                sel = np.flatnonzero(flag < 50)
                probe.add_input(batch_of([pk[sel]]))
            maps, rows, pays = [], [], []
            while True:
                m, r, cols, fin = probe.get_output(40000, [0])
                maps.append(np.asarray(m))
                rows.append(np.asarray(r))
                pays.append(np.asarray(cols[0][0]))
                if fin:
                    break
            m = np.concatenate(maps)
            out.append((m if sel is None else sel[m], np.concatenate(rows), np.concatenate(pays)))
        res[impl.__name__] = out
    for g, e in zip(res[vx.__name__], res[oracle.__name__]):
        for a, b_ in zip(g, e):
            assert len(a) == len(b_) and (a == b_).all()
        assert len(g[0]) > 5000


def test_queued_no_more_input_and_output_pages(oracle, vx):
    """vx355_agg_no_more_input_async / vx355_agg_get_output_async (ABI 7): noMoreInput and several pages of
    output queued behind the batches, none of the calls waits; 'done' fires on the worker thread per page
    with (status, rows, finished); vx355_agg_output_result hands each page out once, and only after its
    ticket completed. Same groups in the same order as the synchronous path and the oracle."""
    import threading
    rng = np.random.default_rng(14)
    batches = _batches(rng, 60, 5000)
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_MIN, 2, abi.BIGINT), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    exp, _ = run_agg(oracle, batches, [0], [abi.BIGINT], aggs, max_rows=4096)
    op = vx.Aggregation([0], [abi.BIGINT], aggs)
    for b in batches:
        op.add_input_async(b)
    assert op.no_more_input_async() == 61
    fired, all_done = [], threading.Event()

    def done(status, n, fin):
        fired.append((status, n, fin))
        if fin:
            all_done.set()

    pages = [op.get_output_async(200, done) for _ in range(3)]      # 500 groups: 200 + 200 + 100
    assert [t for t, _ in pages] == [62, 63, 64]
    early = pages[2][0]
    if op.poll()[1] < early:
        with pytest.raises(vx.Vx355Error) as e:
            op.output_result(early, pages[2][1])                     # not there yet: said so, nothing handed out
        assert e.value.status == abi.EINVAL
    assert all_done.wait(60)
    assert fired == [(0, 200, False), (0, 200, False), (0, 100, True)]
    assert op.poll() == (64, 64)
    cols = None
    for ticket, out in pages:
        page, n, fin = op.output_result(ticket, out)
        assert (n, fin) == ((200, False) if ticket < 64 else (100, True))
        cols = page if cols is None else [(np.concatenate([a[0], b[0]]), np.concatenate([a[1], b[1]]))
                                          for a, b in zip(cols, page)]
    assert_columns_equal(cols, exp, op.kinds, what="queued output pages")
    with pytest.raises(vx.Vx355Error):
        op.output_result(pages[0][0], pages[0][1])                   # handed out once
    # a page queued behind a failed batch is skipped; its callback and its result carry the failure
    op = vx.Aggregation([0], [abi.BIGINT], [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_SUM, 2, abi.BIGINT)])
    op.add_input_async(batch_of([rng.integers(0, 5, 10).astype(np.int64)]))   # the plan reads columns 1 and 2
    seen, skipped = [], threading.Event()
    op.no_more_input_async()
    ticket, out = op.get_output_async(100, lambda status, n, fin: (seen.append(status), skipped.set()))
    assert skipped.wait(60) and seen == [abi.EINVAL]
    with pytest.raises(vx.Vx355Error) as e:
        op.output_result(ticket, out)
    assert e.value.status == abi.EINVAL


def test_probe_output_pages_queued_behind_the_batch(oracle, vx):
    """vx355_join_probe_get_output_async / _output_result (ABI 7): the probe batch and the pages of its output
    queue up on the worker; nobody waits until the last callback has fired. A LEFT join (misses keep their probe
    row) and then the unmatched build rows of a RIGHT join (build_side = 1): the same lists as the synchronous
    calls give."""
    import threading
    rng = np.random.default_rng(18)
    nb = 50_000
    bk = rng.permutation(1_000_000)[:nb].astype(np.int64)
    pay = rng.integers(0, 1 << 30, nb).astype(np.int64)
    pk = rng.integers(0, 1_000_000, 200_000).astype(np.int64)
    pk[::7] = bk[rng.integers(0, nb, len(pk[::7]))]
    for kind in (abi.JOIN_LEFT, abi.JOIN_RIGHT):
        lists = {}
        for how in ("sync", "queued"):
            b = vx.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], kind)
            b.add_input(batch_of([bk, pay]))
            probe = vx.JoinProbe(b.finish(), [0], kind)
            hb = batch_of([pk])
            got = []
            if how == "sync":
                probe.add_input(hb)
                while True:
                    m, r, cols, fin = probe.get_output(30000, [0])
                    got.append((np.asarray(m), np.asarray(r), np.asarray(cols[0][0]), np.asarray(cols[0][1])))
                    if fin:
                        break
                if kind == abi.JOIN_RIGHT:
                    while True:
                        r, cols, fin = probe.get_build_side_output(30000, [0])
                        got.append((np.full(len(r), -1, np.int32), np.asarray(r), np.asarray(cols[0][0]),
                                    np.asarray(cols[0][1])))
                        if fin:
                            break
            else:
                ticket = probe.add_input_async(hb)
                for build_side in ([False, True] if kind == abi.JOIN_RIGHT else [False]):
                    while True:
                        fired = threading.Event()
                        seen = []
                        t, page = probe.get_output_async(30000, [0], lambda s, n, f: (seen.append((s, n, f)), fired.set()),
                                                         build_side=build_side)
                        assert t > ticket
                        assert fired.wait(60) and seen[0][0] == 0
                        m, r, cols, fin = probe.output_result(t, page)
                        assert (len(m), fin) == (seen[0][1], seen[0][2])
                        got.append((np.asarray(m) if not build_side else np.full(len(r), -1, np.int32), np.asarray(r),
                                    np.asarray(cols[0][0]), np.asarray(cols[0][1])))
                        if fin:
                            break
                with pytest.raises(vx.Vx355Error):
                    probe.output_result(t, page)    # handed out once
            lists[how] = [np.concatenate([g[i] for g in got]) for i in range(4)]
        for a, b2 in zip(lists["sync"], lists["queued"]):
            assert len(a) == len(b2) > 0 and (a == b2).all()
        assert len(lists["sync"][0]) >= len(pk) if kind == abi.JOIN_LEFT else True
