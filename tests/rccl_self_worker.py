"""Worker of test_gpu_dist_abi.py::test_rccl_entry_points_run_on_one_gpu (own process: creating an
RCCL communicator changes how the rest of a process runs, and a hang must not take pytest along).

VX355_COMM_FORCE_RCCL=1: a ONE-rank communicator really calls ncclCommInitRank, and every collective
of the library goes through RCCL with the rank's own slice as a send / recv pair to itself inside
the ncclGroupStart / ncclGroupEnd bracket - the code path N > 1 ranks take (GroupGuard, sendBytes'
256 MiB cutting, the all-gather of counts, the exchange edge's payload stream and receive slots),
executed on a 1-GPU box. Prints one JSON line of checks."""
import ctypes as C
import faulthandler
import json
import os
import sys

faulthandler.enable()   # a crash inside RCCL leaves a Python traceback on stderr

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["VX355_COMM_FORCE_RCCL"] = "1"
if "--with-torch" in sys.argv:
    # the arrangement of bench.py: torch first, so that libvx355 runs on torch's bundled HIP runtime and
    # loads the librccl that ships with it (a different RCCL build than /opt/rocm's)
    sys.argv.remove("--with-torch")
    import torch  # noqa: F401

from velox_amd import abi, ops as vx  # noqa: E402
from gpu_util import batch_of  # noqa: E402


def dev_alloc(nbytes):
    p = vx.lib().vx355_device_malloc(nbytes)
    assert p, "device allocation failed"
    return C.c_void_p(p)


def h2d(p, arr):
    vx._check(vx.lib().vx355_memcpy_h2d(p, arr.ctypes.data, arr.nbytes))


def d2h(p, n, dtype):
    out = np.empty(n, dtype=dtype)
    vx._check(vx.lib().vx355_memcpy_d2h(out.ctypes.data, p, out.nbytes))
    return out


def main():
    vx.init(0)
    checks = {}
    def progress(what):
        print("rccl_self_worker:", what, file=sys.stderr, flush=True)
    progress("creating the communicator")
    comm = vx.Comm(vx.Comm.unique_id(), 1, 0)
    checks["comm_info"] = list(comm.info())
    steps = set(sys.argv[1:]) or {"counts", "columns", "all_gather", "all_gather_large", "all_gather_v", "merge", "edge"}
    if "counts" in steps:
        progress("all-gather of counts")
        checks["counts"] = comm.exchange_counts([12345])
    rng = np.random.default_rng(9)
    # grouped send / recv to self: a 4-byte and an 8-byte column; the 8-byte one is 320 MiB, so
    # sendBytes / recvBytes cut it into two messages
    n = 40 << 20
    a = rng.integers(-2 ** 62, 2 ** 62, n).astype(np.int64)
    b = rng.integers(-2 ** 31, 2 ** 31 - 1, n).astype(np.int32)
    sa, sb, ra, rb = dev_alloc(a.nbytes), dev_alloc(b.nbytes), dev_alloc(a.nbytes), dev_alloc(b.nbytes)
    h2d(sa, a)
    h2d(sb, b)
    zeros = np.zeros(n, dtype=np.int64)
    if "columns" in steps:
        progress("grouped send / recv to self")
        comm.exchange_columns([sa.value, sb.value], [8, 4], [n], [n], [ra.value, rb.value])
        checks["columns_8_byte_320MiB"] = bool((d2h(ra, n, np.int64) == a).all())
        checks["columns_4_byte"] = bool((d2h(rb, n, np.int32) == b).all())
    # all-gather (one ncclAllGather) and the cut form above 256 MiB
    small = 1 << 20
    if "all_gather" in steps:
        progress("all-gather")
        h2d(ra, zeros)
        comm.all_gather(sa.value, ra.value, small)
        checks["all_gather"] = bool((d2h(ra, small // 8, np.int64) == a[: small // 8]).all())
    if "all_gather_large" in steps:
        progress("all-gather, 320 MiB")
        h2d(ra, zeros)
        comm.all_gather(sa.value, ra.value, a.nbytes)
        checks["all_gather_320MiB"] = bool((d2h(ra, n, np.int64) == a).all())
    if "all_gather_v" in steps:
        progress("all-gather of unequal blocks")
        h2d(rb, zeros[: n // 2])
        comm.all_gather_v(sb.value, [b.nbytes], rb.value)
        checks["all_gather_v"] = bool((d2h(rb, n, np.int32) == b).all())
    if "merge" in steps:
        # vx355_agg_merge_partials through the same communicator: partial groups -> one PrestoPage (avg as
        # ROW(DOUBLE, BIGINT)) -> counts + all-gather of the pages over RCCL -> FINAL; equal to one SINGLE
        # aggregation of the same rows (dyadic doubles: exact)
        progress("merge of partials")
        from velox_amd import dist as vdist
        m = 200_000
        key = rng.integers(0, 5000, m).astype(np.int64)
        big = rng.integers(2 ** 44, 2 ** 46, m).astype(np.int64)
        x = rng.integers(-1000, 1000, m).astype(np.float64) / 8
        raw = [(abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_AVG, 2, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT),
               (abi.AGG_MAX, 2, abi.DOUBLE)]
        batch = batch_of([key, big, x], [None, None, rng.random(m) > 0.1])
        single = vx.Aggregation([0], [abi.BIGINT], raw, abi.STEP_SINGLE)
        single.add_input(batch)
        single.no_more_input()
        want = vx.collect_output(single, 4096)
        part = vx.Aggregation([0], [abi.BIGINT], raw, abi.STEP_PARTIAL)
        part.add_input(batch)
        part.no_more_input()
        fin = vx.merge_partials(comm, part, [0], [abi.BIGINT], vdist.final_aggs_for(raw, 1))
        got = vx.collect_output(fin, 4096)
        same = len(got) == len(want)
        for (gv, gn), (wv, wn) in zip(got, want):
            gv, wv = np.asarray(gv), np.asarray(wv)
            gn = np.ones(len(gv), dtype=bool) if gn is None else np.asarray(gn, dtype=bool)
            wn = np.ones(len(wv), dtype=bool) if wn is None else np.asarray(wn, dtype=bool)
            same = same and len(gv) == len(wv) and bool((gn == wn).all()) and bool((gv[gn] == wv[wn]).all())
        checks["merge_partials"] = bool(same)
        del fin, part, single
    if "edge" not in steps:
        del comm
        print(json.dumps(checks), flush=True)
        return
    # the exchange edge: hash + partition + grouping, counts, payload on the edge's own stream
    progress("exchange edge")
    ex = vx.Exchange(comm, [abi.BIGINT, abi.DOUBLE], [0])
    sent = []
    for rows in (100_000, 0, 3_000_001):
        k = rng.integers(-2 ** 40, 2 ** 40, rows).astype(np.int64)
        sent.append((k, rng.random(rows)))
    ex.send(batch_of(list(sent[0])))
    ex.send(vx.to_device(batch_of(list(sent[1]))))
    ok = True
    for i in range(3):
        cols, rows = ex.receive()
        k, v = sent[i]
        ok = ok and rows == len(k)
        gk = np.empty(rows, dtype=np.int64)
        gv = np.empty(rows, dtype=np.float64)
        if rows:
            vx._check(vx.lib().vx355_memcpy_d2h(gk.ctypes.data, cols[0].values, gk.nbytes))
            vx._check(vx.lib().vx355_memcpy_d2h(gv.ctypes.data, cols[1].values, gv.nbytes))
        ok = ok and bool((gk == k).all()) and bool((gv == v).all())   # one destination: the rows keep their order
        if i == 0:
            ex.send(batch_of(list(sent[2])))
    checks["exchange_edge"] = ok
    del ex
    del comm
    print(json.dumps(checks), flush=True)


if __name__ == "__main__":
    main()
