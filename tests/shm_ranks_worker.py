"""One rank of the library's own exchange with SEVERAL ranks on ONE GPU (VX355_COMM_TRANSPORT=shm:
velox_amd/csrc/shm_transport.hip serves the exchange's transport table through host shared memory,
because RCCL refuses two ranks per device). Started world times by tests/test_gpu_dist_abi.py:

    python tests/shm_ranks_worker.py RANK WORLD ID_FILE

1. everything velox_amd/commcheck.py checks (counts, grouped column slices, both all-gathers, a
   300 MiB slice across the 256 MiB message cut, vx355_join_repartition with four pipelined chunks,
   vx355_agg_merge_partials), then
2. the same two plan fragments with UNEVEN and EMPTY shards against the CPU oracle: a repartitioned
   join whose build rows all start on rank 0's neighbours (rank 0 holds none, the last rank holds no
   probe rows) - every joined (fact key, dim payload) pair must be the oracle's -, and a partial ->
   final aggregation in which one rank contributes an empty partial result."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def shard_sizes(world, total, empty_rank):
    """uneven split of 'total' rows: rank r gets a share proportional to r + 1, 'empty_rank' none"""
    weights = np.array([0 if r == empty_rank else r + 1 for r in range(world)], dtype=np.float64)
    sizes = np.floor(total * weights / weights.sum()).astype(np.int64)
    sizes[int(np.argmax(weights))] += total - int(sizes.sum())
    return sizes.tolist()


def uneven_join(ops, oracle, comm, rank, world):
    from velox_amd import abi
    nd, nf = 30_000, 250_000
    rng = np.random.default_rng(4242)            # the same on every rank: each one knows the whole input
    dim_k = (rng.permutation(4 * nd)[:nd].astype(np.int64) * 104729) % (1 << 44)
    dim_a = rng.integers(0, 1 << 40, nd).astype(np.int64)
    fact_k = np.where(rng.random(nf) < 0.8, dim_k[rng.integers(0, nd, nf)], rng.integers(0, 1 << 44, nf)).astype(np.int64)
    fact_m = np.arange(nf, dtype=np.float64)
    bsz, psz = shard_sizes(world, nd, 0), shard_sizes(world, nf, world - 1)
    b0, p0 = int(sum(bsz[:rank])), int(sum(psz[:rank]))
    bk, ba = dim_k[b0:b0 + bsz[rank]], dim_a[b0:b0 + bsz[rank]]
    pk, pm = fact_k[p0:p0 + psz[rank]], fact_m[p0:p0 + psz[rank]]
    # the oracle's answer for the WHOLE join: payload per fact row (or miss)
    ob = oracle.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER)
    ob.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, dim_k), abi.HostColumn(abi.BIGINT, dim_a)]))
    op = oracle.JoinProbe(ob.finish(), [0], abi.JOIN_INNER)
    op.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, fact_k)]))
    want = {}
    while True:
        m, r, cols, fin = op.get_output(1 << 20, [0])
        for row, pay in zip(np.asarray(m).tolist(), np.asarray(cols[0][0]).tolist()):
            want[row] = pay
        if fin:
            break
    build = ops.to_device(abi.HostBatch([abi.HostColumn(abi.BIGINT, bk), abi.HostColumn(abi.BIGINT, ba)]))
    probe = ops.to_device(abi.HostBatch([abi.HostColumn(abi.BIGINT, pk), abi.HostColumn(abi.DOUBLE, pm)]))
    cap = nf + 1024
    mapping, brows = ops.DeviceArray(cap, np.int32), ops.DeviceArray(cap, np.int32)
    pay, nulls = ops.DeviceArray(cap, np.int64), ops.DeviceArray(cap // 64 + 1, np.uint64)
    descs = (abi.OutColumn * 1)()
    descs[0].type_kind, descs[0].mem = abi.BIGINT, abi.MEM_DEVICE
    descs[0].values, descs[0].nulls = pay.ptr, nulls.ptr
    got = []   # (global fact row, payload) of every joined row that landed on this rank

    def sink(chunk, received, probe_op):
        # 'received': the fact rows of this chunk that hashed here; column 1 carries the global row number
        batch = received.contents
        rows = int(batch.num_rows)
        marks = np.zeros(rows, dtype=np.float64)
        if rows:
            ops._check(ops.lib().vx355_memcpy_d2h(marks.ctypes.data, batch.cols[1].values, rows * 8))
        while True:
            n, fin = probe_op.get_output_device(cap, mapping.ptr, brows.ptr, descs, [0])
            if n:
                got.append(np.stack([marks[mapping.to_host(n)].astype(np.int64), pay.to_host(n)], axis=1))
            if fin:
                break
    ops.join_repartition(comm, ([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER), build,
                         ([0], abi.JOIN_INNER), probe, 3, sink)
    mine = np.concatenate(got) if got else np.zeros((0, 2), dtype=np.int64)
    for row, payload in mine.tolist():
        if want.get(row) != payload:
            print(f"shm ranks: fact row {row} joined payload {payload}, the oracle says {want.get(row)}", file=sys.stderr)
            return 21
    # every oracle pair exactly once over all ranks
    counts = ops.DeviceArray(np.array([len(mine), len(set(mine[:, 0].tolist()))], dtype=np.int64))
    allc = ops.DeviceArray(2 * world, np.int64)
    comm.all_gather(counts.ptr, allc.ptr, 16)
    tot = allc.to_host().reshape(world, 2).sum(axis=0)
    if int(tot[0]) != len(want) or int(tot[1]) != len(want):
        print(f"shm ranks: {tot.tolist()} joined rows over all ranks, the oracle has {len(want)}", file=sys.stderr)
        return 22
    return 0


def uneven_merge(ops, oracle, comm, rank, world):
    from velox_amd import abi
    from velox_amd import dist as vdist
    total = 120_000
    rng = np.random.default_rng(777)
    k = rng.integers(0, 5000, total).astype(np.int64)
    v = rng.integers(-(1 << 30), 1 << 30, total).astype(np.int64)
    d = rng.integers(0, 1 << 20, total).astype(np.float64) / 16
    sizes = shard_sizes(world, total, 1 if world > 1 else -1)     # rank 1 aggregates nothing
    at = int(sum(sizes[:rank]))
    sl = slice(at, at + sizes[rank])
    aggs = [(abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_COUNT_STAR, -1, abi.BIGINT), (abi.AGG_AVG, 2, abi.DOUBLE),
            (abi.AGG_MIN, 1, abi.BIGINT), (abi.AGG_MAX, 2, abi.DOUBLE)]
    partial = ops.HashAggregation([0], [abi.BIGINT], aggs, abi.STEP_PARTIAL)
    if sizes[rank]:
        partial.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, k[sl]), abi.HostColumn(abi.BIGINT, v[sl]),
                                         abi.HostColumn(abi.DOUBLE, d[sl])]))
    partial.no_more_input()
    final = ops.merge_partials(comm, partial, [0], [abi.BIGINT], vdist.final_aggs_for(aggs, 1))
    out = ops.collect_output(final, 8192)
    single = oracle.Aggregation([0], [abi.BIGINT], aggs)
    single.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, k), abi.HostColumn(abi.BIGINT, v),
                                    abi.HostColumn(abi.DOUBLE, d)]))
    single.no_more_input()
    exp = oracle.collect_output(single, 8192)
    order_g, order_e = np.argsort(np.asarray(out[0][0])), np.argsort(np.asarray(exp[0][0]))
    for c, (g, e) in enumerate(zip(out, exp)):
        gv, ev = np.asarray(g[0])[order_g], np.asarray(e[0])[order_e]
        if len(gv) != len(ev) or not (gv == ev).all() or not (np.asarray(g[1])[order_g] == np.asarray(e[1])[order_e]).all():
            print(f"shm ranks: merged aggregation column {c} differs from the oracle's SINGLE aggregation", file=sys.stderr)
            return 23
    return 0


def second_phase(argv):
    """(own communicator: commcheck released its own)"""
    import time
    rank, world, id_file = int(argv[0]), int(argv[1]), argv[2] + ".uneven"
    import oracle_lib
    from velox_amd import ops
    oracle_lib.lib()
    if rank == 0:
        uid = ops.Comm.unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(id_file + ".tmp", id_file)
    else:
        deadline = time.time() + 60
        while not os.path.exists(id_file):
            if time.time() > deadline:
                return 30
            time.sleep(0.05)
        with open(id_file, "rb") as f:
            uid = f.read()
    comm = ops.Comm(uid, world, rank)
    if comm.info() != (world, rank, 0):
        return 31
    rc = uneven_join(ops, oracle_lib, comm, rank, world) or uneven_merge(ops, oracle_lib, comm, rank, world)
    del comm
    if rc == 0:
        print(f"shm ranks: rank {rank}/{world} uneven and empty shards match the oracle")
    return rc


if __name__ == "__main__":
    os.environ["VX355_COMM_TRANSPORT"] = "shm"
    from velox_amd import commcheck
    code = commcheck.main([sys.argv[1], sys.argv[2], "0", sys.argv[3]])
    if code == 0:
        code = second_phase(sys.argv[1:])
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code)   # (no tear-down: see velox_amd/commcheck.py)
