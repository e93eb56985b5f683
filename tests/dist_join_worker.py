"""One rank of the repartitioned join on CPU/gloo for tests/test_dist_gloo.py.
The oracle stands in for the GPU kernels (hash, partition, local join); the
exchange code under test is velox_amd/dist.py."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def shard(rank, world, n_dim=3000, n_fact=20000):
    """dim: unique pk, row-range partitioned; fact: fk uniform over all keys."""
    rng = np.random.default_rng(500 + rank)
    pk = (np.arange(rank * n_dim, (rank + 1) * n_dim, dtype=np.int64) * 7919) % (1 << 40)
    a = rng.integers(0, 1 << 30, n_dim).astype(np.int64)
    all_keys = (np.arange(0, world * n_dim, dtype=np.int64) * 7919) % (1 << 40)
    fk = all_keys[rng.integers(0, world * n_dim, n_fact)]
    fk[::17] = -5 - rank  # keys without a partner
    m = rng.random(n_fact)
    return (pk, a), (fk, m)


class OracleBackend:
    def __init__(self, torch, oracle, abi):
        self.torch, self.oracle, self.abi = torch, oracle, abi

    def partitions(self, key, world):
        from velox_amd import dist as vdist
        abi = self.abi
        k = key.numpy()
        batch = abi.HostBatch([abi.HostColumn(abi.BIGINT, k)], len(k))
        h = self.oracle.hash_columns(batch, [0])
        kind, kw = vdist.partition_spec(world)
        return self.torch.from_numpy(self.oracle.partition(h, kind, **kw).astype(np.int32))

    def scatter(self, parts, world, cols):
        p = parts.numpy()
        order = np.argsort(p, kind="stable")
        counts = np.bincount(p, minlength=world).astype(np.int64)
        return [c[self.torch.from_numpy(order)] for c in cols], counts

    def build(self, build_cols):
        abi, oracle = self.abi, self.oracle
        b = oracle.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER)
        b.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, build_cols[0].numpy()),
                                   abi.HostColumn(abi.BIGINT, build_cols[1].numpy())]))
        return b.finish()

    def join(self, build_cols, probe_cols):
        return self.probe(self.build(build_cols), probe_cols)

    def probe(self, table, probe_cols):
        abi, oracle = self.abi, self.oracle
        p = oracle.JoinProbe(table, [0], abi.JOIN_INNER)
        fk, m = probe_cols[0].numpy(), probe_cols[1].numpy()
        p.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, fk)], len(fk)))
        out = []
        while True:
            mapping, rows, cols, fin = p.get_output(4096)
            for i in range(len(mapping)):
                out.append((int(fk[mapping[i]]), float(m[mapping[i]]), int(cols[0][0][i])))
            if fin:
                break
        return out


def run(rank, world, port, out_dir, chunks=0, max_message=0):
    import torch
    import torch.distributed as dist
    import oracle_lib
    from velox_amd import abi
    from velox_amd import dist as vdist
    if max_message:
        vdist.MAX_MESSAGE_BYTES = max_message
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    (pk, a), (fk, m) = shard(rank, world)
    backend = OracleBackend(torch, oracle_lib, abi)
    build_cols = [torch.from_numpy(pk), torch.from_numpy(a)]
    probe_cols = [torch.from_numpy(fk), torch.from_numpy(m)]
    if chunks:
        # chunked exchange overlapped with the next chunk's partitioning (async all-to-all)
        per_chunk, _ = vdist.repartitioned_join_pipelined(backend, dist, torch, build_cols, probe_cols, chunks)
        rows = [r for _, out in per_chunk for r in out]
    else:
        rows = vdist.repartitioned_join(backend, dist, torch, build_cols, probe_cols)
    np.save(os.path.join(out_dir, f"join_rank{rank}.npy"), np.array(rows, dtype=np.float64).reshape(-1, 3))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else 0,
        int(sys.argv[6]) if len(sys.argv) > 6 else 0)
