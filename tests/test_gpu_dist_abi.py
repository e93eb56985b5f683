"""The multi-GPU orchestration behind the C ABI at the world size a 1-GPU box offers:
vx355_exchange_* (PartitionedOutput -> Exchange edge), vx355_join_repartition,
vx355_agg_merge_partials, the communicator self-check and bench.py's own launcher."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from velox_amd import abi
from velox_amd import dist as vdist

from gpu_util import assert_columns_equal, batch_of

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _device_values(vx, column, rows, dtype):
    out = np.empty(rows, dtype=dtype)
    if rows:
        vx._check(vx.lib().vx355_memcpy_d2h(out.ctypes.data, column.values, out.nbytes))
    return out


def test_exchange_edge_round_trip_and_pipelining(vx):
    """send / receive on a one-rank communicator: every row comes back, in order, for host and
    device batches; two sends may be in flight, a third is refused until one is received."""
    comm = vx.Comm(vx.Comm.unique_id(), 1, 0)
    assert comm.info() == (1, 0, 0)
    ex = vx.Exchange(comm, [abi.BIGINT, abi.DOUBLE, abi.INTEGER], [0])
    rng = np.random.default_rng(5)
    sent = []
    for n in (1000, 0, 70001):
        k = rng.integers(-2 ** 40, 2 ** 40, n).astype(np.int64)
        v = rng.random(n)
        d = rng.integers(0, 1000, n).astype(np.int32)
        sent.append((k, v, d))
    ex.send(batch_of(list(sent[0])))
    ex.send(vx.to_device(batch_of(list(sent[1]))))
    with pytest.raises(vx.Vx355Error):
        ex.send(batch_of(list(sent[2])))
    for i in range(3):
        cols, rows = ex.receive()
        k, v, d = sent[i]
        assert rows == len(k)
        assert (_device_values(vx, cols[0], rows, np.int64) == k).all()
        assert (_device_values(vx, cols[1], rows, np.float64) == v).all()
        assert (_device_values(vx, cols[2], rows, np.int32) == d).all()
        if i == 0:
            ex.send(batch_of(list(sent[2])))
    with pytest.raises(vx.Vx355Error):
        ex.receive()
    # nulls and encodings do not travel on this edge
    with pytest.raises(vx.Vx355Error) as e:
        ex.send(batch_of([sent[0][0], sent[0][1], sent[0][2]], [None, np.arange(1000) % 3 != 0, None]))
    assert e.value.status == abi.EUNSUPPORTED


@pytest.mark.parametrize("world", [2, 3, 4, 8, 7])
def test_exchange_destinations_equal_the_cpu_hash_partition(oracle, vx, world):
    """What the edge groups by at N > 1 (the fused hash + partition kernel) against the oracle's
    VectorHasher::hash and HashPartitionFunction::partition: top hash bits for powers of two,
    hash % world otherwise; one and two key columns, host and device batches."""
    rng = np.random.default_rng(world)
    n = 50_001
    k1 = rng.integers(-2 ** 50, 2 ** 50, n).astype(np.int64)
    k2 = rng.integers(0, 1000, n).astype(np.int32)
    v = rng.random(n)
    comm = vx.Comm(vx.Comm.unique_id(), 1, 0)
    for key_cols in ([0], [0, 1]):
        ex = vx.Exchange(comm, [abi.BIGINT, abi.INTEGER, abi.DOUBLE], key_cols)
        batch = batch_of([k1, k2, v])
        hashes = oracle.hash_columns(batch, key_cols)
        kind, kw = vdist.partition_spec(world)
        want = oracle.partition(hashes, kind, **kw)
        assert (ex.destinations(batch, world) == want).all()
        assert (ex.destinations(vx.to_device(batch), world) == want).all()
        assert len(np.unique(want)) == world


@pytest.mark.parametrize("regroup", [False, True])
def test_join_repartition_matches_the_oracle_join(oracle, vx, regroup, monkeypatch):
    """vx355_join_repartition (exchange + build, then the probe side in pipelined chunks through
    the sink) against the oracle's join of the same rows. regroup: every chunk reaches the sink
    regrouped by slice of the join table (vx355_join_probe_add_input_regrouped, forced here on a
    small table) - 'received' is then the regrouped batch and the mappings number its rows."""
    if regroup:
        monkeypatch.setenv("VX355_JOIN_REGROUP", "1")
        monkeypatch.setenv("VX355_JOIN_SLICE_BYTES", "65536")
        monkeypatch.setenv("VX355_JOIN_ARRAY_MAX", "0")
    rng = np.random.default_rng(52)
    nd, nf = 40000, 300001
    pk = rng.permutation(1 << 20)[:nd].astype(np.int64)
    a = rng.integers(0, 1 << 40, nd).astype(np.int64)
    fk = np.where(rng.random(nf) < 0.85, pk[rng.integers(0, nd, nf)], -7).astype(np.int64)
    m = rng.random(nf)
    comm = vx.Comm(vx.Comm.unique_id(), 1, 0)
    build = vx.to_device(batch_of([pk, a]))
    probe = vx.to_device(batch_of([fk, m]))
    got_keys, got_payload, got_m, seen_chunks = [], [], [], []

    def sink(chunk, received, probe_op):
        seen_chunks.append(chunk)
        rows = received.contents.num_rows
        keys = _device_values(vx, received.contents.cols[0], rows, np.int64)
        ms = _device_values(vx, received.contents.cols[1], rows, np.float64)
        cap = 50000
        mapping = vx.DeviceArray(cap, np.int32)
        brows = vx.DeviceArray(cap, np.int32)
        pay = vx.DeviceArray(cap, np.int64)
        nulls = vx.DeviceArray(cap // 64 + 1, np.uint64)
        descs = (abi.OutColumn * 1)()
        descs[0].type_kind, descs[0].mem = abi.BIGINT, abi.MEM_DEVICE
        descs[0].values, descs[0].nulls = pay.ptr, nulls.ptr
        while True:
            n, fin = probe_op.get_output_device(cap, mapping.ptr, brows.ptr, descs, [0])
            mp = mapping.to_host(n)
            got_keys.append(keys[mp])
            got_m.append(ms[mp])
            got_payload.append(pay.to_host(n).copy())
            if fin:
                break
    vx.profile_reset()
    vx.profile_enable(True)
    table = vx.join_repartition(comm, ([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER), build,
                                ([0], abi.JOIN_INNER), probe, 3, sink)
    vx.profile_enable(False)
    assert seen_chunks == [0, 1, 2] and table.stats().num_rows == nd
    b = oracle.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER)
    b.add_input(batch_of([pk, a]))
    t = b.finish()
    p = oracle.JoinProbe(t, [0], abi.JOIN_INNER)
    p.add_input(batch_of([fk, m]))
    want = []
    while True:
        mp, r, cols, fin = p.get_output(1 << 20)
        want += list(zip(fk[mp].tolist(), m[mp].tolist(), np.asarray(cols[0][0]).tolist()))
        if fin:
            break
    got = list(zip(np.concatenate(got_keys).tolist(), np.concatenate(got_m).tolist(),
                   np.concatenate(got_payload).tolist()))
    if regroup:
        assert sorted(got) == sorted(want) and len(got) > 100000
        assert "k_grp_scatter" in vx.profile()
    else:
        # one rank, chunks in row order: the same rows in the same order as the single join
        assert got == want and len(got) > 100000

    def failing(chunk, received, probe_op):
        raise RuntimeError("sink gave up")
    with pytest.raises(RuntimeError, match="sink gave up"):
        vx.join_repartition(comm, ([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER), build,
                            ([0], abi.JOIN_INNER), probe, 2, failing)


@pytest.mark.parametrize("groups", [5, 3000])
def test_merge_partials_equals_a_single_aggregation(oracle, vx, groups):
    """partial -> PrestoPage -> all-gather -> final inside the library (vx355_agg_merge_partials)
    equals one SINGLE aggregation: integers beyond 2^53, INT64 min / max, avg, a string key of
    more than 12 bytes (non-inline views through the page path), null keys and null inputs."""
    rng = np.random.default_rng(groups)
    n = 60000
    names = [b"group %06d with a long name" % i if i % 3 else b"g%d" % i for i in range(groups)]
    gid = rng.integers(0, groups, n)
    key = [names[i] for i in gid]
    key_valid = rng.random(n) > 0.02
    big = rng.integers(2 ** 44, 2 ** 46, n).astype(np.int64)   # group sums beyond 2^53, inside int64
    x = rng.integers(-1000, 1000, n).astype(np.float64) / 8
    x_valid = rng.random(n) > 0.1
    raw = [(abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_MIN, 1, abi.BIGINT), (abi.AGG_MAX, 1, abi.BIGINT),
           (abi.AGG_AVG, 2, abi.DOUBLE), (abi.AGG_COUNT, 2, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT),
           (abi.AGG_SUM, 2, abi.DOUBLE)]
    batch = batch_of([key, big, x], [key_valid, None, x_valid])
    single = oracle.Aggregation([0], [abi.VARCHAR], raw, abi.STEP_SINGLE)
    single.add_input(batch)
    single.no_more_input()
    exp = oracle.collect_output(single, 4096)
    part = vx.Aggregation([0], [abi.VARCHAR], raw, abi.STEP_PARTIAL)
    part.add_input(batch)
    part.no_more_input()
    comm = vx.Comm(vx.Comm.unique_id(), 1, 0)
    fin = vx.merge_partials(comm, part, [0], [abi.VARCHAR], vdist.final_aggs_for(raw, 1))
    got = vx.collect_output(fin, 4096)
    kinds = [abi.VARCHAR, abi.BIGINT, abi.BIGINT, abi.BIGINT, abi.DOUBLE, abi.BIGINT, abi.BIGINT, abi.DOUBLE]
    assert_columns_equal(got, exp, kinds, float_ulps=0, what=f"{groups} groups")   # dyadic doubles: exact


def test_commcheck_world_size_one(tmp_path):
    """The communicator self-check bench.py runs before it commits to the in-library exchange."""
    r = subprocess.run([sys.executable, "-m", "velox_amd.commcheck", "0", "1", "0", str(tmp_path / "id")], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "ok" in r.stdout


def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` without a launcher: bench.py starts the ranks itself; on a 1-GPU
    box they share the GPU (VX355_BENCH_SHARE_GPU) and exchange through the LIBRARY (vx355_agg_merge_partials
    over the shared-memory transport): the line must say 2 and must not report a downgraded exchange."""
    env = dict(os.environ, VX355_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "200000", "--steps", "2",
                        "--warmup", "1", "--no-traffic", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["rows_per_gpu"] == 200000
    assert line["value"] > 0 and line["scaling"] == "weak"
    assert line["strong_scaling"]["rows_total"] == 200000
    assert "exchange_downgraded" not in line and "shared-memory transport" in line["config"]["exchange"]


def test_bench_c5_through_the_library_exchange():
    """--workload c5 at N = 1 runs vx355_join_repartition end to end (the exchange is a device copy)."""
    detail = os.path.join(tempfile.mkdtemp(prefix="vx355_bench_"), "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c5", "--rows", "2000000", "--steps",
                        "2", "--warmup", "1", "--no-traffic", "--no-cpu-baseline", "--detail", detail],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.rstrip("\n").splitlines()[-1]
    assert len(last) < 4096                                       # the compact line the driver parses
    line = json.loads(last)
    assert line["n_gpus"] == 1 and "libvx355" in line["config"]["exchange"]
    assert line["detail"] == detail and line["result_check_ok"] is True
    full = json.load(open(detail))                                # everything else is in the detail file
    assert full["workload_info"]["matches_on_rank0"] == 2000000   # every fact row finds its dim row


@pytest.mark.parametrize("with_torch", [False, True])
def test_rccl_entry_points_run_on_one_gpu(with_torch):
    """VX355_COMM_FORCE_RCCL=1 (tests/rccl_self_worker.py, own process): ncclCommInitRank, the counts
    all-gather, grouped ncclSend / ncclRecv to self with a 320 MiB column cut at 256 MiB, both
    all-gather forms, vx355_agg_merge_partials (avg travels as ROW(DOUBLE, BIGINT) in the page) and the
    exchange edge's payload stream really execute; every payload comes back intact. The N > 1 code path no longer meets RCCL for the first time on the 8-GPU node.
    with_torch: torch is imported first, as in bench.py - the library then runs on torch's bundled
    HIP runtime and the RCCL next to it (another build than /opt/rocm's)."""
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "rccl_self_worker.py")] + (["--with-torch"] if with_torch else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    checks = json.loads([line for line in r.stdout.splitlines() if line.startswith("{")][-1])
    assert checks.pop("comm_info") == [1, 0, 0]
    assert checks.pop("counts") == [12345]
    assert checks and all(checks.values()), checks


@pytest.mark.parametrize("world", [2, 3])
def test_library_exchange_with_several_ranks_on_one_gpu(world, tmp_path):
    """The library's own exchange between DIFFERENT ranks, on the one GPU a test box has:
    VX355_COMM_TRANSPORT=shm puts a host shared-memory transport under the exchange's transport
    table (RCCL refuses two ranks per device), everything above it is the code the 8-GPU node runs:
    vx355_exchange_counts / _columns with real peers, both all-gathers, a 300 MiB slice across the
    message cut, vx355_join_repartition (chunk-count agreement, three receive slots, the payload
    stream) and vx355_agg_merge_partials - first with even shards (velox_amd/commcheck.py), then with
    uneven and empty shards against the CPU oracle (tests/shm_ranks_worker.py)."""
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["VX355_COMM_TRANSPORT"] = "shm"
    id_file = str(tmp_path / "comm_id")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shm_ranks_worker.py"), str(r), str(world), id_file],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=900))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: rc {p.returncode}\n{out[-1500:]}\n{err[-3000:]}"
        assert f"commcheck rank {r}/{world} on device 0: ok" in out
        assert f"rank {r}/{world} uneven and empty shards match the oracle" in out


@pytest.mark.parametrize("world", [2, 3])
def test_one_process_drives_every_rank(world):
    """vx355_comm_create_all (one process drives the node, one Driver thread per rank):
    the repartitioned join and the merged aggregation with uneven and empty shards against the oracle,
    every rank a thread of ONE process with its own communicator, execution contexts and streams
    (tests/one_process_ranks_worker.py; the ranks share GPU 0 through the shared-memory transport)."""
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "one_process_ranks_worker.py"), str(world)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert f"one process, {world} ranks: join and merged aggregation match the oracle on every rank" in r.stdout
