"""Self-check of libvx355's own RCCL communicator at world size > 1, run as one short-lived
process per rank BEFORE a multi-GPU run commits to the in-library exchange:

    python -m velox_amd.commcheck RANK WORLD DEVICE ID_FILE

Rank 0 writes the communicator id to ID_FILE, the others wait for it; every rank creates its
vx355_comm, compares what the communicator reports (vx355_comm_info: ncclCommCount /
ncclCommUserRank / ncclCommCuDevice) with what it was told, and runs each collective the hot
path uses once on a few bytes (slice sizes, grouped send / recv of column slices, all-gather,
all-gather of unequal blocks). Exit code 0 = usable. The caller (bench.py) runs it under a
timeout: a hang or a failure makes the run fall back to torch.distributed for the exchange and
say so in its JSON line, instead of hanging the benchmark.
"""
import os
import sys
import time

import numpy as np


def main(argv):
    rank, world, device, id_file = int(argv[0]), int(argv[1]), int(argv[2]), argv[3]
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
    del comm
    print(f"commcheck rank {rank}/{world} on device {device}: ok")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
