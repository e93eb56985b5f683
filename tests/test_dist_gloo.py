"""The N > 1 path on CPU: two processes over gloo run partial aggregation on
their own shards, all-gather the partial rows (velox_amd/dist.py) and run the
final step; the result must equal a single aggregation over both shards,
including the first-seen group order."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from velox_amd import abi

import dist_worker

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_partial_final_matches_single_process(oracle, tmp_path):
    _two_rank_partial_final(oracle, tmp_path, None)


@pytest.mark.gpu
def test_two_rank_partial_final_with_the_library_as_the_operator(oracle, tmp_path):
    """The same two-rank plan with libvx355 (not the oracle) as the operator on both ranks, which share GPU 0:
    partial aggregation on the GPU, groups through torch.distributed (gloo), final aggregation on the GPU,
    against the oracle's SINGLE aggregation of all rows - bit for bit, group order included."""
    _two_rank_partial_final(oracle, tmp_path, "vx")


def _two_rank_partial_final(oracle, tmp_path, impl):
    port = _free_port()
    world = 2
    env = dict(os.environ)
    if impl:
        env["VX355_DIST_IMPL"] = impl
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), str(world),
                               str(port), str(tmp_path)], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    key_types = [abi.VARCHAR, abi.VARCHAR]
    single = oracle.Aggregation([0, 1], key_types, dist_worker.raw_aggs(abi, impl == "vx"), abi.STEP_SINGLE)
    for r in range(world):
        single.add_input(dist_worker.batch_for(abi, r))
    single.no_more_input()
    exp = oracle.collect_output(single, 4096)
    import pickle
    for r in range(world):
        with open(os.path.join(tmp_path, f"rank{r}.pkl"), "rb") as f:
            got = pickle.load(f)
        assert len(got) == len(exp)
        for (gv, gvalid), (ev, evalid) in zip(got, exp):
            evalid = np.asarray(evalid).tolist()
            assert gvalid == evalid
            ev = list(ev) if isinstance(ev, list) else np.asarray(ev).tolist()
            # dyadic doubles and integers (beyond 2^53 included): every bit must survive the merge
            assert [g for g, ok in zip(gv, evalid) if ok] == [e for e, ok in zip(ev, evalid) if ok]
    # the plan whose partial rows travel as PrestoPages (900 groups, long string keys)
    single = oracle.Aggregation([0], [abi.VARCHAR], dist_worker.wide_aggs(abi), abi.STEP_SINGLE)
    for r in range(world):
        single.add_input(dist_worker.wide_batch(abi, r))
    single.no_more_input()
    exp = oracle.collect_output(single, 4096)
    for r in range(world):
        with open(os.path.join(tmp_path, f"wide_rank{r}.pkl"), "rb") as f:
            got = pickle.load(f)
        for (gv, gvalid), (ev, evalid) in zip(got, exp):
            evalid = np.asarray(evalid).tolist()
            assert gvalid == evalid
            ev = list(ev) if isinstance(ev, list) else np.asarray(ev).tolist()
            assert [g for g, ok in zip(gv, evalid) if ok] == [e for e, ok in zip(ev, evalid) if ok]


def test_gather_encoding_round_trip():
    from velox_amd import dist as vdist
    i64 = np.iinfo(np.int64)
    cols = [([b"A", b"", b"RETURN", None, b"twelve bytes", b"12345678"], np.array([True, True, True, False, True, True])),
            (np.array([1.5, -2.0, 0.0, 7.0, np.nextafter(1.0, 2.0), -0.0]), np.array([True, False, True, True, True, True])),
            (np.array([3, 0, 2 ** 40, -5, i64.max, i64.min], dtype=np.int64), np.ones(6, dtype=bool)),
            (np.array([2 ** 53 + 1, -(2 ** 53) - 1, 2 ** 62 + 12345, 1, 2, 3], dtype=np.int64), np.ones(6, dtype=bool)),
            (np.array([1.25, 3.0, -1.0, 0.5, 2.0 ** -20, 1e30], dtype=np.float32), np.ones(6, dtype=bool))]
    kinds = [abi.VARCHAR, abi.DOUBLE, abi.BIGINT, abi.BIGINT, abi.REAL]
    batch = vdist.decode_partials(vdist.encode_partial(cols, kinds), kinds)
    assert batch.num_rows == 6
    from velox_amd.abi import view_to_bytes
    back = [view_to_bytes(batch.columns[0].values[i]) for i in (0, 1, 2, 4, 5)]
    assert back == [b"A", b"", b"RETURN", b"twelve bytes", b"12345678"] and not batch.columns[0].valid[3]
    assert (batch.columns[1].values.view(np.int64) == cols[1][0].view(np.int64)).all()   # -0.0 and 1 + ulp included
    assert (batch.columns[1].valid == cols[1][1]).all()
    assert (batch.columns[2].values == cols[2][0]).all()   # INT64 min / max
    assert (batch.columns[3].values == cols[3][0]).all()   # beyond 2^53: no float64 detour
    assert (batch.columns[4].values == cols[4][0]).all()
    with pytest.raises(ValueError):
        vdist.encode_partial([([b"thirteen bytes"], np.array([True]))], [abi.VARCHAR])


import pytest


@pytest.mark.parametrize("world,chunks,max_message", [(2, 0, 0), (3, 0, 0), (2, 3, 0), (3, 5, 0), (2, 0, 4096), (3, 2, 4096)])
def test_repartitioned_join_over_gloo_matches_single_process(world, chunks, max_message, tmp_path):
    """BASELINE config 5 on CPU: hash-partition both sides, one all-to-all per
    column (velox_amd/dist.py), local joins; the union over ranks must equal the
    single-process join of the concatenated inputs. world = 3 exercises the
    hash % n flavour, world = 2 the bit-range flavour; chunks > 0 the pipelined form
    (probe side in chunks, asynchronous all-to-all overlapped with the next chunk's
    partitioning); max_message > 0 forces the exchange to cut its slices into several rounds
    (what it does above 256 MiB per message on the GPU)."""
    import dist_join_worker
    import pandas as pd
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_join_worker.py"), str(r), str(world),
                               str(port), str(tmp_path), str(chunks), str(max_message)]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    got = np.concatenate([np.load(os.path.join(tmp_path, f"join_rank{r}.npy")) for r in range(world)])
    dims, facts = zip(*[dist_join_worker.shard(r, world) for r in range(world)])
    dim = pd.DataFrame({"k": np.concatenate([d[0] for d in dims]), "a": np.concatenate([d[1] for d in dims])})
    fact = pd.DataFrame({"k": np.concatenate([f[0] for f in facts]), "m": np.concatenate([f[1] for f in facts])})
    want = fact.merge(dim, on="k")
    assert len(got) == len(want) and len(got) > 0
    g = sorted(map(tuple, got.tolist()))
    w = sorted(zip(want["k"].astype(float), want["m"], want["a"].astype(float)))
    assert g == w
    # every key ended up on exactly one rank
    per_rank_keys = [set(np.load(os.path.join(tmp_path, f"join_rank{r}.npy"))[:, 0].tolist()) for r in range(world)]
    for i in range(world):
        for j in range(i + 1, world):
            assert not (per_rank_keys[i] & per_rank_keys[j])


def test_partial_gather_small_and_large_group_counts():
    """all_gather_partials: one fixed-size collective up to 64 groups per rank, a second,
    exactly sized one above (driven here by a 1-rank stand-in for torch.distributed)."""
    import torch
    from velox_amd import dist as vdist

    class OneRank:
        calls = 0

        def get_world_size(self):
            return 1

        def all_gather(self, out, t):
            OneRank.calls += 1
            out[0].copy_(t)
    kinds = [abi.BIGINT, abi.DOUBLE]
    for groups, expect_calls in ((5, 1), (64, 1), (65, 2), (3000, 2)):
        cols = [(np.arange(groups, dtype=np.int64) * 3, np.ones(groups, dtype=bool)),
                (np.arange(groups) / 8.0, np.arange(groups) % 5 != 0)]
        OneRank.calls = 0
        batch = vdist.all_gather_partials(OneRank(), torch, cols, kinds)
        assert OneRank.calls == expect_calls and batch.num_rows == groups
        assert (batch.columns[0].values == cols[0][0]).all()
        assert (batch.columns[1].values == cols[1][0]).all() and (batch.columns[1].valid == cols[1][1]).all()
