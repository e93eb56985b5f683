"""Self-check of libvx355's own RCCL communicator at world size > 1, run as one short-lived
process per rank BEFORE a multi-GPU run commits to the in-library exchange:

    python -m velox_amd.commcheck RANK WORLD DEVICE ID_FILE

Rank 0 writes the communicator id to ID_FILE, the others wait for it; every rank creates its
vx355_comm, compares what the communicator reports (vx355_comm_info: ncclCommCount /
ncclCommUserRank / ncclCommCuDevice) with what it was told, and runs each collective the hot
path uses once on a few bytes (slice sizes, grouped send / recv of column slices, all-gather,
all-gather of unequal blocks), then one 300 MiB slice (the library cuts messages at 256 MiB), a small
repartitioned join through vx355_join_repartition with four pipelined chunks and a merge of partial
aggregations through vx355_agg_merge_partials, each verified. Exit code 0 = usable. The caller (bench.py) runs it under a
timeout: a hang or a failure makes the run fall back to torch.distributed for the exchange and
say so in its JSON line, instead of hanging the benchmark.
"""
import os
import sys
import time

import numpy as np


def main(argv):
    rank, world, device, id_file = int(argv[0]), int(argv[1]), int(argv[2]), argv[3]
    # bench.py has torch loaded before libvx355: the library then runs on torch's bundled HIP runtime
    # and loads the librccl next to it. The check must exercise THAT pair, not /opt/rocm's.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    from velox_amd import ops
    ops.init(device)
    if rank == 0:
        uid = ops.Comm.unique_id()
        tmp = id_file + ".tmp"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, id_file)
    else:
        deadline = time.time() + 60
        while not os.path.exists(id_file):
            if time.time() > deadline:
                print("commcheck: no communicator id from rank 0", file=sys.stderr)
                return 3
            time.sleep(0.05)
        with open(id_file, "rb") as f:
            uid = f.read()
    comm = ops.Comm(uid, world, rank)
    got = comm.info()
    if got != (world, rank, device):
        print(f"commcheck: communicator reports {got}, expected {(world, rank, device)}", file=sys.stderr)
        return 4
    # slice sizes: rank r sends r + p + 1 rows to rank p
    send = [rank + p + 1 for p in range(world)]
    recv = comm.exchange_counts(send)
    if recv != [s + rank + 1 for s in range(world)]:
        print(f"commcheck: exchange_counts returned {recv}", file=sys.stderr)
        return 5
    src = np.concatenate([np.full(send[p], rank * 1000 + p, dtype=np.int64) for p in range(world)])
    d_src, d_dst = ops.DeviceArray(src), ops.DeviceArray(int(sum(recv)), np.int64)
    comm.exchange_columns([d_src.ptr], [8], send, recv, [d_dst.ptr])
    want = np.concatenate([np.full(recv[s], s * 1000 + rank, dtype=np.int64) for s in range(world)])
    if not (d_dst.to_host() == want).all():
        print("commcheck: exchange_columns delivered the wrong slices", file=sys.stderr)
        return 6
    one = ops.DeviceArray(np.array([rank + 7], dtype=np.int64))
    everyone = ops.DeviceArray(world, np.int64)
    comm.all_gather(one.ptr, everyone.ptr, 8)
    if everyone.to_host().tolist() != [r + 7 for r in range(world)]:
        print("commcheck: all_gather delivered the wrong blocks", file=sys.stderr)
        return 7
    sizes = [8 * (r + 1) for r in range(world)]
    mine = ops.DeviceArray(np.full(rank + 1, rank, dtype=np.int64))
    blocks = ops.DeviceArray(sum(sizes) // 8, np.int64)
    comm.all_gather_v(mine.ptr, sizes, blocks.ptr)
    if blocks.to_host().tolist() != [r for r in range(world) for _ in range(r + 1)]:
        print("commcheck: all_gather_v delivered the wrong blocks", file=sys.stderr)
        return 8
    rc = check_large_message(ops, comm, rank, world) or check_join(ops, comm, rank, world) or \
        check_merge(ops, comm, rank, world)
    if rc:
        return rc
    del comm
    print(f"commcheck rank {rank}/{world} on device {device}: ok")
    return 0


def check_large_message(ops, comm, rank, world):
    """One slice above the 256 MiB message cut (exchange.hip: kMaxMessageBytes) to the next rank:
    the pieces must arrive whole and in order."""
    n = (300 << 20) // 8
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    send = [n if p == nxt else 0 for p in range(world)]
    recv = comm.exchange_counts(send)
    if recv != [n if s == prv else 0 for s in range(world)]:
        print(f"commcheck: exchange_counts (large) returned {recv}", file=sys.stderr)
        return 9
    src = np.arange(n, dtype=np.int64) * 3 + rank
    d_src, d_dst = ops.DeviceArray(src), ops.DeviceArray(n, np.int64)
    comm.exchange_columns([d_src.ptr], [8], send, recv, [d_dst.ptr])
    got = d_dst.to_host()
    if not (got == np.arange(n, dtype=np.int64) * 3 + prv).all():
        bad = int(np.flatnonzero(got != np.arange(n, dtype=np.int64) * 3 + prv)[0])
        print(f"commcheck: a {n * 8 >> 20} MiB message arrived damaged from element {bad}", file=sys.stderr)
        return 10
    return 0


def check_join(ops, comm, rank, world):
    """vx355_join_repartition with four pipelined chunks (exchange edges, three receive slots, the
    payload stream): every fact row must find its dim row wherever it lives; checked by count and
    by a checksum of the joined payload over all ranks."""
    from velox_amd import abi
    nd, nf = 20_000, 400_000
    rng = np.random.default_rng(100 + rank)
    pk = (np.arange(rank * nd, (rank + 1) * nd, dtype=np.int64) * 7919) % (1 << 45)
    a = rng.integers(0, 1 << 40, nd).astype(np.int64)
    idx = rng.integers(0, world * nd, nf).astype(np.int64)
    fk = (idx * 7919) % (1 << 45)
    m = rng.random(nf)
    everyone_a = [np.random.default_rng(100 + r).integers(0, 1 << 40, nd).astype(np.int64) for r in range(world)]
    want_sum = int(np.concatenate(everyone_a)[idx].astype(np.uint64).sum(dtype=np.uint64))
    hb = abi.HostBatch([abi.HostColumn(abi.BIGINT, pk), abi.HostColumn(abi.BIGINT, a)])
    hp = abi.HostBatch([abi.HostColumn(abi.BIGINT, fk), abi.HostColumn(abi.DOUBLE, m)])
    build, probe = ops.to_device(hb), ops.to_device(hp)
    cap = nf + 1024
    mapping, brows = ops.DeviceArray(cap, np.int32), ops.DeviceArray(cap, np.int32)
    pay, nulls = ops.DeviceArray(cap, np.int64), ops.DeviceArray(cap // 64 + 1, np.uint64)
    descs = (abi.OutColumn * 1)()
    descs[0].type_kind, descs[0].mem = abi.BIGINT, abi.MEM_DEVICE
    descs[0].values, descs[0].nulls = pay.ptr, nulls.ptr
    total, got_sum = [0], [0]

    def sink(chunk, received, probe_op):
        while True:
            n, fin = probe_op.get_output_device(cap, mapping.ptr, brows.ptr, descs, [0])
            total[0] += n
            got_sum[0] = (got_sum[0] + int(pay.to_host(n).astype(np.uint64).sum(dtype=np.uint64))) & ((1 << 64) - 1)
            if fin:
                break
    ops.join_repartition(comm, ([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER), build,
                         ([0], abi.JOIN_INNER), probe, 4, sink)
    # totals over all ranks: rows, payload checksum, expected checksum
    mine = ops.DeviceArray(np.array([total[0], got_sum[0], want_sum], dtype=np.uint64))
    allv = ops.DeviceArray(3 * world, np.uint64)
    comm.all_gather(mine.ptr, allv.ptr, 24)
    v = allv.to_host().reshape(world, 3)
    rows, got, want = int(v[:, 0].sum()), int(v[:, 1].sum(dtype=np.uint64)), int(v[:, 2].sum(dtype=np.uint64))
    if rows != world * nf or got != want:
        print(f"commcheck: repartitioned join produced {rows} rows (want {world * nf}), checksum {got} vs {want}",
              file=sys.stderr)
        return 11
    return 0


def check_merge(ops, comm, rank, world):
    """vx355_agg_merge_partials: per-rank partial sums / counts of 1000 groups, merged on every
    rank; group g's count must be what all ranks together contributed."""
    from velox_amd import abi
    n = 50_000
    rng = np.random.default_rng(200 + rank)
    k = rng.integers(0, 1000, n).astype(np.int64)
    v = rng.integers(0, 1 << 20, n).astype(np.int64)
    aggs = [(abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    partial = ops.HashAggregation([0], [abi.BIGINT], aggs, abi.STEP_PARTIAL)
    partial.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, k), abi.HostColumn(abi.BIGINT, v)]))
    partial.no_more_input()
    from velox_amd import dist as vdist
    final = ops.merge_partials(comm, partial, [0], [abi.BIGINT], vdist.final_aggs_for(aggs, 1))
    out = ops.collect_output(final, 4096)
    keys, sums, counts = (np.asarray(c[0]) for c in out)
    want_sum, want_cnt = np.zeros(1000, dtype=np.int64), np.zeros(1000, dtype=np.int64)
    for r in range(world):
        g = np.random.default_rng(200 + r)
        kr = g.integers(0, 1000, n).astype(np.int64)
        vr = g.integers(0, 1 << 20, n).astype(np.int64)
        np.add.at(want_sum, kr, vr)
        np.add.at(want_cnt, kr, 1)
    live = want_cnt > 0
    if len(keys) != int(live.sum()) or not (sums == want_sum[keys]).all() or not (counts == want_cnt[keys]).all():
        print("commcheck: merged partial aggregation differs from the sum over the ranks", file=sys.stderr)
        return 12
    return 0


if __name__ == "__main__":
    # A throw-away process: leave without the interpreter's and the runtimes' tear-down (several processes
    # letting go of one GPU at the same moment have crashed there, after every check had passed).
    code = main(sys.argv[1:])
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code)
