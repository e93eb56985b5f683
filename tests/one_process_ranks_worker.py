"""ONE process drives all ranks ("one process drives all 8 GPUs"): vx355_comm_create_all,
then one Driver thread per rank runs the two distributed plan fragments - vx355_join_repartition and
vx355_agg_merge_partials, uneven and empty shards - against the CPU oracle. On a box with one GPU the
ranks share device 0 and the exchange's transport table is served by the host shared-memory transport
(VX355_COMM_TRANSPORT=shm: RCCL refuses two ranks per device); everything above the table - per-rank
execution contexts and streams inside one process, counts first, grouped send / recv per peer, chunk
agreement, the three receive slots - is what a node's worth of GPUs runs.

    python tests/one_process_ranks_worker.py WORLD
"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(world):
    os.environ["VX355_COMM_TRANSPORT"] = "shm"
    import oracle_lib
    import shm_ranks_worker as cases
    from velox_amd import ops
    oracle_lib.lib()
    ops.init(0)
    comms = ops.Comm.create_all([0] * world)
    assert [c.info() for c in comms] == [(world, r, 0) for r in range(world)], [c.info() for c in comms]
    codes = [None] * world

    def driver(rank):
        try:
            ops.set_device(0)
            codes[rank] = (cases.uneven_join(ops, oracle_lib, comms[rank], rank, world) or
                           cases.uneven_merge(ops, oracle_lib, comms[rank], rank, world))
        except BaseException as e:  # noqa: BLE001 - reported as the rank's failure
            print(f"one process: rank {rank} raised {e!r}", file=sys.stderr)
            codes[rank] = 40
    threads = [threading.Thread(target=driver, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    if any(t.is_alive() for t in threads):
        print("one process: a rank's thread did not finish", file=sys.stderr)
        return 41
    if any(codes):
        return max(c or 0 for c in codes)
    print(f"one process, {world} ranks: join and merged aggregation match the oracle on every rank")
    return 0


if __name__ == "__main__":
    code = main(int(sys.argv[1]))
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code)   # (no tear-down: see velox_amd/commcheck.py)
