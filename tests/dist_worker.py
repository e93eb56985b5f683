"""Worker for tests/test_dist_gloo.py: one rank of the sharded aggregation
(partial step per rank -> all-gather -> final step) over gloo. The operators are the
oracle's on a box without a GPU (the -m "not gpu" suite: the exchange logic of
velox_amd/dist.py with world_size 2) and the LIBRARY's when VX355_DIST_IMPL=vx (the
-m gpu suite: both ranks share GPU 0, the groups travel through torch.distributed)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def shard(rank, n=20000):
    rng = np.random.default_rng(100 + rank)
    flags = [bytes([c]) for c in rng.choice(list(b"ANR"), n)]
    status = [bytes([c]) for c in rng.choice(list(b"FO"), n)]
    qty = rng.integers(1, 51, n).astype(np.float64)
    price = rng.integers(0, 1 << 20, n).astype(np.float64) / 64
    valid = rng.random(n) > 0.1
    # integers far above 2^53: the N > 1 merge must keep every bit (sum / min / max of BIGINT)
    big = rng.integers(1 << 54, 1 << 56, n).astype(np.int64) * rng.choice([-1, 1], n)
    return flags, status, qty, price, valid, big


def raw_aggs(abi, device=None):
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_AVG, 3, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT),
            (abi.AGG_MIN, 3, abi.DOUBLE), (abi.AGG_COUNT, 3, abi.DOUBLE), (abi.AGG_MAX, 2, abi.DOUBLE),
            (abi.AGG_SUM, 4, abi.BIGINT), (abi.AGG_MIN, 4, abi.BIGINT), (abi.AGG_MAX, 4, abi.BIGINT)]
    if device is None:
        device = os.environ.get("VX355_DIST_IMPL") == "vx"
    # one device table holds 16 accumulators (DOUBLE sums and BIGINT sums own two words each): the
    # library refuses the nine-aggregate plan at create time, as it would tell the adapter to
    return aggs[:7] if device else aggs


def batch_for(abi, rank):
    flags, status, qty, price, valid, big = shard(rank)
    return abi.HostBatch([abi.HostColumn(abi.VARCHAR, flags), abi.HostColumn(abi.VARCHAR, status),
                          abi.HostColumn(abi.DOUBLE, qty), abi.HostColumn(abi.DOUBLE, price, valid),
                          abi.HostColumn(abi.BIGINT, big)])


def wide_batch(abi, rank, n=30000):
    """Many groups and keys beyond the inline string limit: the merge goes through PrestoPages."""
    rng = np.random.default_rng(500 + rank)
    ids = rng.integers(0, 900, n)
    keys = [b"a grouping key well beyond twelve bytes #%03d" % i for i in ids]
    x = rng.integers(-(1 << 40), 1 << 40, n).astype(np.int64)
    return abi.HostBatch([abi.HostColumn(abi.VARCHAR, keys), abi.HostColumn(abi.BIGINT, x, rng.random(n) > 0.05)])


def wide_aggs(abi):
    return [(abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_COUNT, 1, abi.BIGINT), (abi.AGG_MAX, 1, abi.BIGINT)]


def run(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from velox_amd import abi
    from velox_amd import dist as vdist
    if os.environ.get("VX355_DIST_IMPL") == "vx":
        from velox_amd import ops as oracle_lib   # the product library is the operator on both ranks
        oracle_lib.init(0)
    else:
        import oracle_lib
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    key_types = [abi.VARCHAR, abi.VARCHAR]
    op = oracle_lib.Aggregation([0, 1], key_types, raw_aggs(abi), abi.STEP_PARTIAL)
    op.add_input(batch_for(abi, rank))
    op.no_more_input()
    part = oracle_lib.collect_output(op, 4096)
    merged = vdist.merge_partials(oracle_lib, dist, torch, part, key_types, raw_aggs(abi), None)
    # every rank holds the same final result
    import pickle
    with open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb") as f:
        pickle.dump([(np.asarray(col[0]).tolist() if not isinstance(col[0], list) else list(col[0]),
                      np.asarray(col[1]).tolist()) for col in merged], f)
    # second plan: ~900 groups per rank, long string keys -> page transport
    op = oracle_lib.Aggregation([0], [abi.VARCHAR], wide_aggs(abi), abi.STEP_PARTIAL)
    op.add_input(wide_batch(abi, rank))
    op.no_more_input()
    part = oracle_lib.collect_output(op, 4096)
    merged = vdist.merge_partials(oracle_lib, dist, torch, part, [abi.VARCHAR], wide_aggs(abi), None)
    with open(os.path.join(out_dir, f"wide_rank{rank}.pkl"), "wb") as f:
        pickle.dump([(np.asarray(col[0]).tolist() if not isinstance(col[0], list) else list(col[0]),
                      np.asarray(col[1]).tolist()) for col in merged], f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
