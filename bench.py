#!/usr/bin/env python
"""bench.py — one "step" = one pass of the hot path over one resident batch.

Default workload (BASELINE.json configs[1]): TPC-H Q1 at SF100 — 600 037 902
synthetic lineitem rows per GPU, resident in HBM in Velox's FlatVector layout
(two 16-byte StringView key columns, four DOUBLE columns, one DATE column =
68 B/row) -> FilterProject (l_shipdate <= 1998-09-02; disc_price, charge) ->
HashAggregation (2 keys, 8 aggregates), exactly the reference's plan
(exec/tests/utils/TpchQueryBuilder.cpp:203-252). Everything between the
resident input columns and the final 4-6 group rows on the host is inside the
timed region.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Other workloads (not the headline line): --workload c1 | c4 | q3.
Rank 0 writes the full result object to --detail (bench_detail.json) and to stderr on a
line tagged "BENCH_DETAIL " and then, as the LAST stdout line, the compact JSON line
(< 4 KB: contract keys + roofline + cpu_baseline + reduced secondary blocks).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from velox_amd import abi, ops  # noqa: E402
from velox_amd import dist as vdist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)

SHIP_LO, SHIP_HI = 8036, 10561     # 1992-01-02 .. 1998-12-01, days since epoch
Q1_CUTOFF = 10471                   # 1998-09-02
STATUS_DATE = 9298                  # 1995-06-17


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (20 timed steps: runs of 5 steps scattered 6.6 - 7.1 ms on one box within a minute, 400 steps give 6.81 - 6.86;
    # profiles/r05_q1_step_count_variance.txt)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="q1", choices=["q1", "q1x4", "c1", "c4", "q3", "q3full", "c5"],
                    help="q1 = the reference's TPC-H Q1 plan (2 keys, 8 aggregates; the headline); q1x4 = BASELINE.json's "
                         "wording of it (4 keys, 6 aggregates)")
    ap.add_argument("--rows", type=int, default=0, help="override the row count (debug)")
    ap.add_argument("--cpu-sample-rows", type=int, default=16_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-mt", action="store_true", help="skip the one-worker-per-core CPU leg")
    ap.add_argument("--c1-stream", action="store_true",
                    help="c1: feed 10 000-row host vectors instead of one resident batch")
    ap.add_argument("--host-stream", action="store_true",
                    help="q1: the scan arrives as 10 000-row HOST vectors through the asynchronous boundary (use --rows)")
    ap.add_argument("--q3-random-probe", action="store_true",
                    help="q3: lineitems in random order instead of dbgen's l_orderkey clustering")
    ap.add_argument("--c4-sparse", action="store_true",
                    help="c4: keys = splitmix64(j), j < distinct (no dense range: open-addressing mode)")
    ap.add_argument("--unfused", action="store_true",
                    help="q1: FilterProject and HashAggregation as two operators")
    ap.add_argument("--c4-unordered", action="store_true",
                    help="c4: VX355_AGG_UNORDERED_OUTPUT (the consumer does not need first-seen group order)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="q1 at N = 1: skip the Q3 SF100 join block (the other half of BASELINE's metric)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the rocprofv3 FETCH_SIZE / WRITE_SIZE passes behind roofline.traffic")
    ap.add_argument("--counter-child", action="store_true",
                    help="(internal) a child run under rocprofv3 --pmc: the timed steps only, no second pass with "
                         "the profiler's events - every dispatch of the run belongs to one of the --steps steps")
    ap.add_argument("--secondary-steps", type=int, default=5)
    ap.add_argument("--secondary", default="",
                    help="comma-separated keys of the secondary blocks to run (default: all of bench.SECONDARY)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every GPU gets the full per-GPU row count (default); strong = the "
                         "N = 1 row count (SF100 for q1) is sharded N ways, BASELINE's 1/2/4/8 metric")
    ap.add_argument("--exchange", default="auto", choices=["auto", "lib", "torch"],
                    help="N > 1 data path: lib = libvx355's own RCCL communicator (vx355_exchange_* / "
                         "vx355_agg_merge_partials), torch = torch.distributed collectives; auto = lib after "
                         "a self-check in child processes (velox_amd/commcheck.py), else torch")
    ap.add_argument("--detail", default="bench_detail.json",
                    help="file for the full result object (every per-kernel time, pass table and note); the last "
                         "stdout line is the compact form, always < 4 KB. '' = do not write a file")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, form the process group and report n_gpus (no GPU work; CPU test of the launcher)")
    return ap.parse_args()


def free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def self_spawn(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves
    (one process per GPU, the same command the driver uses) and pass their output through."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def library_comm(args, torch, dist, rank, world, dev_index, share):
    """The communicator of the in-library exchange, or (None, why not). With --exchange auto the
    communicator is first tried in throw-away child processes under a timeout, so that a RCCL
    problem on this node degrades the run to torch.distributed instead of hanging it."""
    import subprocess
    import tempfile
    if args.exchange == "torch":
        return None, "--exchange torch"
    if share:
        # RCCL refuses two ranks per device: the library's exchange runs over its shared-memory transport
        # (velox_amd/csrc/shm_transport.hip) - the same protocol code above it, a few GB/s below it
        os.environ.setdefault("VX355_COMM_TRANSPORT", "shm")
    ok = 1
    why = ""
    if args.exchange == "auto" and world > 1:
        box = [None]
        if rank == 0:
            box[0] = os.path.join(tempfile.gettempdir(), "vx355_comm_id_%d_%d" % (os.getpid(), free_port()))
        dist.broadcast_object_list(box, src=0)
        try:
            r = subprocess.run([sys.executable, "-X", "faulthandler", "-m", "velox_amd.commcheck", str(rank), str(world),
                                str(dev_index), box[0]], cwd=ROOT, capture_output=True, text=True, timeout=150)
            # (the child prints its verdict after the last check: a crash while the process winds down - seen
            # with several processes letting go of one GPU - does not undo it)
            if r.returncode != 0 and (": ok" not in r.stdout):
                ok, why = 0, "commcheck rank %d: rc %d %s" % (rank, r.returncode, r.stderr.strip()[-1500:])
        except subprocess.TimeoutExpired:
            ok, why = 0, "commcheck rank %d timed out" % rank
        flag = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        reasons = [None] * world
        dist.all_gather_object(reasons, why)
        if rank == 0 and os.path.exists(box[0]):
            os.unlink(box[0])
        if int(flag.item()) == 0:
            return None, "; ".join(x for x in reasons if x) or "commcheck failed"
    # (one rank: no RCCL at all, not even the id - see vx355_comm_create)
    forced = os.environ.get("VX355_COMM_FORCE_RCCL", "0") not in ("", "0")   # one rank, but through RCCL all the same
    uid = [(ops.Comm.unique_id() if (world > 1 or forced) else bytes(128)) if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(uid, src=0)
    comm = ops.Comm(uid[0], world, rank)
    got = comm.info()
    if got != (world, rank, dev_index):
        raise SystemExit(f"RCCL reports (world, rank, device) = {got}, the launcher said {(world, rank, dev_index)}")
    if os.environ.get("VX355_COMM_TRANSPORT") == "shm":
        return comm, "libvx355 communicator over its shared-memory transport (ranks share one GPU; vx355_comm_create)"
    return comm, "libvx355 RCCL communicator (vx355_comm_create)"


def dcol(kind, tensor, indices=None, base_size=0):
    """torch tensor in HBM -> DeviceColumn aliasing it."""
    if indices is None:
        return ops.DeviceColumn.from_ptr(kind, tensor.data_ptr(), tensor.shape[0])
    return ops.DeviceColumn.from_ptr(kind, tensor.data_ptr(), 0, None, abi.DICTIONARY,
                                     indices.data_ptr(), base_size)


class DevBatch:
    """vx355_batch over DeviceColumns with an explicit row count."""

    def __init__(self, cols, num_rows):
        self.cols = cols
        self.num_rows = num_rows
        self._descs = (abi.Column * len(cols))(*[c.descriptor() for c in cols])
        self.batch = abi.Batch(num_rows, len(cols), self._descs)

    def ref(self):
        return C.byref(self.batch)


def raw_host_column(kind, values):
    """FLAT host column over a numpy buffer that already has Velox's layout (StringViews as (n, 16)
    uint8 rows): no per-value conversion."""
    col = abi.HostColumn.__new__(abi.HostColumn)
    col.kind, col.encoding, col.keep = kind, abi.FLAT, []
    col.values = np.ascontiguousarray(values)
    col.base_size = col.num_rows = len(col.values)
    col.indices = col.valid = col.nulls = None
    return col


def stream_host_batches(op, batches):
    """Feeds an operator the way a Velox Driver does - many small HOST vectors - through the
    asynchronous boundary (queue, parallel ingest into pinned chunks, one upload + launch per chunk);
    returns the milliseconds the calling thread spent queueing."""
    t0 = time.perf_counter()
    for b in batches:
        op.add_input_async(b)
    queued = (time.perf_counter() - t0) * 1e3
    op.wait()
    return queued


# ---------------------------------------------------------------- Q1 ----------
def string_views(torch, codes):
    """1-char inline StringViews (type/StringView.h:76-77) from byte codes."""
    sv = torch.zeros((codes.shape[0], 4), dtype=torch.int32, device=codes.device)
    sv[:, 0] = 1
    sv[:, 1] = codes
    return sv


def gen_q1(torch, n, device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def ri(lo, hi, dtype=torch.int32):
        return torch.randint(lo, hi, (n,), dtype=dtype, device=device, generator=g)
    ship = ri(SHIP_LO, SHIP_HI + 1)
    receipt = ship + ri(1, 31)
    ra = ri(0, 2)
    rf = torch.where(receipt > STATUS_DATE, torch.full_like(ship, 78),
                     torch.where(ra == 1, torch.full_like(ship, 82), torch.full_like(ship, 65)))
    ls = torch.where(ship > STATUS_DATE, torch.full_like(ship, 79), torch.full_like(ship, 70))
    del receipt, ra
    cols = {"rf": string_views(torch, rf), "ls": string_views(torch, ls)}
    del rf, ls
    qty = ri(1, 51).to(torch.float64)
    cols["qty"] = qty
    cols["ep"] = qty * (ri(90000, 210001).to(torch.float64) / 100.0)
    cols["disc"] = ri(0, 11).to(torch.float64) / 100.0
    cols["tax"] = ri(0, 9).to(torch.float64) / 100.0
    cols["ship"] = ship
    return cols


Q1_TERMS = [(6, abi.CMP_LE, Q1_CUTOFF)]
Q1_PROJ = [[(3, 1.0, 0.0), (4, -1.0, 1.0)], [(3, 1.0, 0.0), (4, -1.0, 1.0), (5, 1.0, 1.0)]]
# post-FilterProject columns: 0 rf, 1 ls, 2 qty, 3 ep, 4 disc, 5 disc_price, 6 charge
Q1_AGGS = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_SUM, 3, abi.DOUBLE), (abi.AGG_SUM, 5, abi.DOUBLE),
           (abi.AGG_SUM, 6, abi.DOUBLE), (abi.AGG_AVG, 2, abi.DOUBLE), (abi.AGG_AVG, 3, abi.DOUBLE),
           (abi.AGG_AVG, 4, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
Q1_KEYS = ([0, 1], [abi.VARCHAR, abi.VARCHAR])


class Q1:
    name = "tpch_q1_sf100"
    bytes_per_row = 68          # Velox layout of the 7 scanned columns
    fused = True

    @property
    def agg_bytes_per_row(self):
        # fused: the 7 scan columns (68 B); unfused, at the HashAggregation
        # boundary: 2 x 16 + 5 x 8 = 72 B
        return 68 if self.fused else 72

    @property
    def dominant(self):
        return "k_agg_fast"

    def __init__(self, torch, n, device, seed):
        self.torch, self.n = torch, n
        self.c = gen_q1(torch, n, device, seed)
        c = self.c
        disc_col = dcol(abi.DOUBLE, c["disc"])
        null_frac = float(os.environ.get("VX355_Q1_NULLS", "0"))
        if null_frac > 0:
            # --q1 with a nullable l_discount (1 bit per row, 1 = valid): the plan must stay on k_agg_fast
            words = (n + 63) // 64
            bits = torch.ones(words * 64, dtype=torch.bool, device=device)
            bits[:n] = torch.rand(n, device=device, generator=torch.Generator(device=device).manual_seed(seed + 99)) >= null_frac
            weights = (torch.ones(64, dtype=torch.int64, device=device) << torch.arange(64, device=device))
            self.disc_nulls = torch.zeros(words, dtype=torch.int64, device=device)
            for lo in range(0, words, 1 << 22):   # pack in slices: the int64 expansion is 8 bytes per bit
                hi = min(words, lo + (1 << 22))
                self.disc_nulls[lo:hi] = (bits[lo * 64:hi * 64].view(-1, 64).to(torch.int64) * weights).sum(1)
            del bits
            disc_col = ops.DeviceColumn.from_ptr(abi.DOUBLE, c["disc"].data_ptr(), n, self.disc_nulls.data_ptr())
            self.name = "tpch_q1_sf100_nullable_discount"
        self.scan = DevBatch([dcol(abi.VARCHAR, c["rf"]), dcol(abi.VARCHAR, c["ls"]),
                              dcol(abi.DOUBLE, c["qty"]), dcol(abi.DOUBLE, c["ep"]),
                              disc_col, dcol(abi.DOUBLE, c["tax"]),
                              dcol(abi.INTEGER, c["ship"])], n)
        self.idx = torch.empty(n, dtype=torch.int32, device=device)
        self.dp = torch.empty(n, dtype=torch.float64, device=device)
        self.charge = torch.empty(n, dtype=torch.float64, device=device)
        torch.cuda.synchronize()

    FUSED_AGGS = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_SUM, 3, abi.DOUBLE),
                  (abi.AGG_SUM, ops.PROJ(0), abi.DOUBLE), (abi.AGG_SUM, ops.PROJ(1), abi.DOUBLE),
                  (abi.AGG_AVG, 2, abi.DOUBLE), (abi.AGG_AVG, 3, abi.DOUBLE), (abi.AGG_AVG, 4, abi.DOUBLE),
                  (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]

    stream = False   # --host-stream: the scan arrives as 10 000-row HOST vectors (PCIe inclusive)

    def host_batches(self):
        if not hasattr(self, "_host_batches"):
            c = self.c
            host = [(abi.VARCHAR, c["rf"].cpu().numpy().view(np.uint8).reshape(-1, 16)),
                    (abi.VARCHAR, c["ls"].cpu().numpy().view(np.uint8).reshape(-1, 16)),
                    (abi.DOUBLE, c["qty"].cpu().numpy()), (abi.DOUBLE, c["ep"].cpu().numpy()),
                    (abi.DOUBLE, c["disc"].cpu().numpy()), (abi.DOUBLE, c["tax"].cpu().numpy()),
                    (abi.INTEGER, c["ship"].cpu().numpy())]
            self._host_batches = [abi.HostBatch([raw_host_column(k, v[i:i + 10000]) for k, v in host])
                                  for i in range(0, self.n, 10000)]
        return self._host_batches

    def step(self, step_kind=abi.STEP_SINGLE):
        if self.fused:
            # One operator: FilterProject fused into HashAggregation
            # (vx355_agg_set_fused_input); the scan columns are read once.
            op = ops.HashAggregation(Q1_KEYS[0], Q1_KEYS[1], self.FUSED_AGGS, step_kind)
            op.set_fused_input(Q1_TERMS, Q1_PROJ)
            if self.stream:
                self.submit_ms = stream_host_batches(op, self.host_batches())
            else:
                op.add_input(self.scan)
            op.no_more_input()
            self.selected = self.n
            return ops.collect_output(op, 1024)
        return self.step_unfused(step_kind)

    def info(self):
        if self.stream:
            return {"input": "%d x 10 000-row host vectors through vx355_agg_add_input_async (PCIe inclusive)"
                             % len(self.host_batches()),
                    "driver_thread_ms_queueing_the_last_step": round(self.submit_ms, 3),
                    "host_bytes_per_step": self.n * 68}
        return {"input": "one HBM-resident batch"}

    def partial_operator(self):
        """The PARTIAL operator of this rank's shard after noMoreInput (N > 1: vx355_agg_merge_partials drains it)."""
        if not self.fused:
            raise SystemExit("N > 1 with --unfused uses the torch exchange (--exchange torch)")
        op = ops.HashAggregation(Q1_KEYS[0], Q1_KEYS[1], self.FUSED_AGGS, abi.STEP_PARTIAL)
        op.set_fused_input(Q1_TERMS, Q1_PROJ)
        op.add_input(self.scan)
        op.no_more_input()
        self.selected = self.n
        return op

    def shard_view(self, rows):
        """The same workload over the first 'rows' rows of the resident columns (strong scaling)."""
        import copy
        v = copy.copy(self)
        c = {k: t[:rows] for k, t in self.c.items()}
        v.c, v.n = c, rows
        v.scan = DevBatch([dcol(abi.VARCHAR, c["rf"]), dcol(abi.VARCHAR, c["ls"]),
                           dcol(abi.DOUBLE, c["qty"]), dcol(abi.DOUBLE, c["ep"]),
                           dcol(abi.DOUBLE, c["disc"]), dcol(abi.DOUBLE, c["tax"]),
                           dcol(abi.INTEGER, c["ship"])], rows)
        return v

    def step_unfused(self, step_kind=abi.STEP_SINGLE):
        c, n = self.c, self.n
        m = ops.filter_project_device(self.scan, Q1_TERMS, Q1_PROJ, self.idx.data_ptr(),
                                      [self.dp.data_ptr(), self.charge.data_ptr()])
        batch = DevBatch([dcol(abi.VARCHAR, c["rf"], self.idx, n), dcol(abi.VARCHAR, c["ls"], self.idx, n),
                          dcol(abi.DOUBLE, c["qty"], self.idx, n), dcol(abi.DOUBLE, c["ep"], self.idx, n),
                          dcol(abi.DOUBLE, c["disc"], self.idx, n), dcol(abi.DOUBLE, self.dp),
                          dcol(abi.DOUBLE, self.charge)], m)
        op = ops.HashAggregation(Q1_KEYS[0], Q1_KEYS[1], Q1_AGGS, step_kind)
        op.add_input(batch)
        op.no_more_input()
        out = ops.collect_output(op, 1024)
        self.selected = m
        return out

    def rows_per_step(self):
        return self.n

    def host_sample(self, rows):
        rows = min(rows, self.n)
        c = self.c
        return {k: c[k][:rows].cpu().numpy() for k in c}

    def cpu_reference(self, sample, oracle):
        """Velox-algorithm CPU restatement of the same plan on the sample: numpy
        FilterProject (vectorised C) + oracle HashAggregation, one thread."""
        t0 = time.perf_counter()
        sel = np.flatnonzero(sample["ship"] <= Q1_CUTOFF).astype(np.int32)
        dp = (sample["ep"] * (1 - sample["disc"]))[sel]
        charge = dp * (1 + sample["tax"][sel])
        rows = len(sample["ship"])

        def wrap(kind, base):
            col = abi.HostColumn.__new__(abi.HostColumn)
            col.kind, col.encoding, col.keep = kind, abi.DICTIONARY, []
            col.values = np.ascontiguousarray(base)
            col.base_size, col.indices, col.num_rows = rows, sel, len(sel)
            col.valid = col.nulls = None
            return col
        batch = abi.HostBatch([wrap(abi.VARCHAR, sample["rf"].view(np.uint8).reshape(-1, 16)),
                               wrap(abi.VARCHAR, sample["ls"].view(np.uint8).reshape(-1, 16)),
                               wrap(abi.DOUBLE, sample["qty"]), wrap(abi.DOUBLE, sample["ep"]),
                               wrap(abi.DOUBLE, sample["disc"]), abi.HostColumn(abi.DOUBLE, dp),
                               abi.HostColumn(abi.DOUBLE, charge)], len(sel))
        op = oracle.Aggregation(Q1_KEYS[0], Q1_KEYS[1], Q1_AGGS)
        op.add_input(batch)
        op.no_more_input()
        out = oracle.collect_output(op, 1024)
        return out, time.perf_counter() - t0


class Q1FourKeys(Q1):
    """BASELINE.json's literal wording of configs[1]: "8-column scan + filter + 4-key group-by with 6
    aggregates". The reference's own Q1 has 2 keys and 8 aggregates (class Q1, the headline); this is
    the same scan with two more low-cardinality INTEGER keys — l_linenumber (1..7) and a ship-mode
    code (0..6) — and six aggregates: sum(qty), sum(ep), sum(ep * (1 - disc)), avg(qty), avg(disc),
    count(*): 8 columns (2 x 16-byte StringView, 2 x INTEGER, 3 x DOUBLE, 1 x DATE) = 68 B/row,
    4 x 7 x 7 = 196 groups."""
    name = "tpch_q1_sf100_4key_6agg"
    bytes_per_row = 68
    KEYS = ([0, 1, 2, 3], [abi.VARCHAR, abi.VARCHAR, abi.INTEGER, abi.INTEGER])
    TERMS = [(7, abi.CMP_LE, Q1_CUTOFF)]
    PROJ = [[(5, 1.0, 0.0), (6, -1.0, 1.0)]]
    AGGS6 = [(abi.AGG_SUM, 4, abi.DOUBLE), (abi.AGG_SUM, 5, abi.DOUBLE), (abi.AGG_SUM, ops.PROJ(0), abi.DOUBLE),
             (abi.AGG_AVG, 4, abi.DOUBLE), (abi.AGG_AVG, 6, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]

    @property
    def agg_bytes_per_row(self):
        return 68

    def __init__(self, torch, n, device, seed):
        self.torch, self.n = torch, n
        c = gen_q1(torch, n, device, seed)
        del c["tax"]
        g = torch.Generator(device=device)
        g.manual_seed(seed + 7)
        c["lnum"] = torch.randint(1, 8, (n,), dtype=torch.int32, device=device, generator=g)
        c["mode"] = torch.randint(0, 7, (n,), dtype=torch.int32, device=device, generator=g)
        self.c = c
        self.scan = self.make_scan(c, n)
        torch.cuda.synchronize()

    @staticmethod
    def make_scan(c, n):
        return DevBatch([dcol(abi.VARCHAR, c["rf"]), dcol(abi.VARCHAR, c["ls"]), dcol(abi.INTEGER, c["lnum"]),
                         dcol(abi.INTEGER, c["mode"]), dcol(abi.DOUBLE, c["qty"]), dcol(abi.DOUBLE, c["ep"]),
                         dcol(abi.DOUBLE, c["disc"]), dcol(abi.INTEGER, c["ship"])], n)

    def operator(self, step_kind):
        op = ops.HashAggregation(self.KEYS[0], self.KEYS[1], self.AGGS6, step_kind)
        op.set_fused_input(self.TERMS, self.PROJ)
        return op

    def step(self, step_kind=abi.STEP_SINGLE):
        op = self.operator(step_kind)
        op.add_input(self.scan)
        op.no_more_input()
        self.selected = self.n
        return ops.collect_output(op, 1024)

    def partial_operator(self):
        raise SystemExit("the 4-key variant is a single-GPU line")

    def shard_view(self, rows):
        raise SystemExit("the 4-key variant is a single-GPU line")

    def cpu_reference(self, sample, oracle):
        t0 = time.perf_counter()
        sel = np.flatnonzero(sample["ship"] <= Q1_CUTOFF).astype(np.int32)
        dp = (sample["ep"] * (1 - sample["disc"]))[sel]
        rows = len(sample["ship"])

        def wrap(kind, base):
            col = abi.HostColumn.__new__(abi.HostColumn)
            col.kind, col.encoding, col.keep = kind, abi.DICTIONARY, []
            col.values = np.ascontiguousarray(base)
            col.base_size, col.indices, col.num_rows = rows, sel, len(sel)
            col.valid = col.nulls = None
            return col
        # post-FilterProject columns: 0 rf, 1 ls, 2 lnum, 3 mode, 4 qty, 5 ep, 6 disc, 7 disc_price
        batch = abi.HostBatch([wrap(abi.VARCHAR, sample["rf"].view(np.uint8).reshape(-1, 16)),
                               wrap(abi.VARCHAR, sample["ls"].view(np.uint8).reshape(-1, 16)),
                               wrap(abi.INTEGER, sample["lnum"]), wrap(abi.INTEGER, sample["mode"]),
                               wrap(abi.DOUBLE, sample["qty"]), wrap(abi.DOUBLE, sample["ep"]),
                               wrap(abi.DOUBLE, sample["disc"]), abi.HostColumn(abi.DOUBLE, dp)], len(sel))
        aggs = [(abi.AGG_SUM, 4, abi.DOUBLE), (abi.AGG_SUM, 5, abi.DOUBLE), (abi.AGG_SUM, 7, abi.DOUBLE),
                (abi.AGG_AVG, 4, abi.DOUBLE), (abi.AGG_AVG, 6, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
        op = oracle.Aggregation(self.KEYS[0], self.KEYS[1], aggs)
        op.add_input(batch)
        op.no_more_input()
        out = oracle.collect_output(op, 1024)
        return out, time.perf_counter() - t0


# ---------------------------------------------------------------- C1 ----------
class C1:
    """BASELINE configs[0]: SELECT k, sum(v), count(*) GROUP BY k; 10 M rows, 1 K groups."""
    name = "c1_groupby_10m_1k"
    bytes_per_row = 16
    agg_bytes_per_row = 16
    dominant = "k_agg_fast"
    AGGS = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]

    # The input of a step is 160 MB; replayed every step it would sit in the 256 MiB Infinity Cache, whose
    # hits the FETCH_SIZE counter cannot tell from HBM reads (MI355X_MICROARCH.md "Infinity Cache"). The
    # steps therefore rotate through ROTATE distinct inputs (same distribution, other seeds): a working
    # set of 480 MB, so that every step's input comes from HBM. VX355_C1_ROTATE=1 restores the replay.
    ROTATE = int(os.environ.get("VX355_C1_ROTATE", "3"))

    def __init__(self, torch, n, device, seed):
        self.n = n
        self.inputs = []
        null_frac = float(os.environ.get("VX355_C1_NULLS", "0"))
        for i in range(max(1, self.ROTATE)):
            g = torch.Generator(device=device)
            g.manual_seed(seed + 1000 * i)
            k = torch.randint(0, 1000, (n,), dtype=torch.int64, device=device, generator=g)
            v = torch.rand(n, dtype=torch.float64, device=device, generator=g)
            vcol = dcol(abi.DOUBLE, v)
            keep = [k, v]
            if null_frac > 0:
                # the reference's *_halfnull variants (SimpleAggregates.cpp): v carries a null bitmap
                words = (n + 63) // 64
                bits = torch.ones(words * 64, dtype=torch.bool, device=device)
                bits[:n] = torch.rand(n, device=device, generator=g) >= null_frac
                weights = torch.ones(64, dtype=torch.int64, device=device) << torch.arange(64, device=device)
                v_nulls = (bits.view(-1, 64).to(torch.int64) * weights).sum(1)
                vcol = ops.DeviceColumn.from_ptr(abi.DOUBLE, v.data_ptr(), n, v_nulls.data_ptr())
                self.name = "c1_groupby_10m_1k_%d_percent_null_values" % round(null_frac * 100)
                keep += [v_nulls]
                if i == 0:
                    self.v_valid = bits[:n]
            self.inputs.append((DevBatch([dcol(abi.BIGINT, k), vcol], n), keep))
        self.k, self.v = self.inputs[0][1][0], self.inputs[0][1][1]
        self.batch = self.inputs[0][0]
        self.steps_done = 0
        torch.cuda.synchronize()

    stream = False  # --c1-stream: 10 000-row HOST vectors, the way the reference feeds the operator

    def step(self, step_kind=abi.STEP_SINGLE):
        if (type(self) is C1 and not self.stream and step_kind == abi.STEP_SINGLE and
                os.environ.get("VX355_C1_LEAN", "1") != "0"):
            return self.lean_step()
        op = ops.HashAggregation([0], [abi.BIGINT], self.AGGS, step_kind)
        if self.stream:
            if not hasattr(self, "_host_batches"):
                hk, hv = self.k.cpu().numpy(), self.v.cpu().numpy()
                self._host_batches = [abi.HostBatch([abi.HostColumn(abi.BIGINT, hk[i:i + 10000]),
                                                     abi.HostColumn(abi.DOUBLE, hv[i:i + 10000])])
                                      for i in range(0, self.n, 10000)]
            if os.environ.get("VX355_C1_SYNC") == "1":
                for b in self._host_batches:   # synchronous addInput: the calling thread stages every vector itself
                    op.add_input(b)
            else:
                # asynchronous boundary: the Driver thread only queues; copier threads fill pinned chunks,
                # the handle's worker uploads and launches
                self.submit_ms = stream_host_batches(op, self._host_batches)
        else:
            op.add_input(self.inputs[self.steps_done % len(self.inputs)][0])
            self.steps_done += 1
        op.no_more_input()
        return ops.collect_output(op, 4096)

    def lean_step(self):
        """The same step - create, addInput, noMoreInput, getOutput into host buffers, destroy - as five calls of
        the C ABI with the spec and the output buffers built once: what a C++ Driver does per batch. The wrapper
        classes of velox_amd/ops.py (spec arrays, numpy views of the result, __del__) cost 40 - 60 us per step,
        a third of a 0.18-ms step; VX355_C1_LEAN=0 times them as before."""
        L = ops.lib()
        if not hasattr(self, "_lean"):
            probe = ops.HashAggregation([0], [abi.BIGINT], self.AGGS, abi.STEP_SINGLE)
            spec, keep = ops.make_agg_spec([0], [abi.BIGINT], self.AGGS, abi.STEP_SINGLE, False, 0)
            self._lean = (spec, keep, abi.OutBuffers(probe.kinds, 4096), len(probe.kinds))
            del probe
        spec, _keep, out, ncols = self._lean
        h, n, fin = C.c_void_p(), C.c_int32(), C.c_int32()
        ops._check(L.vx355_agg_create(C.byref(spec), C.byref(h)))
        try:
            ops._check(L.vx355_agg_add_input(h, self.inputs[self.steps_done % len(self.inputs)][0].ref()))
            self.steps_done += 1
            ops._check(L.vx355_agg_no_more_input(h))
            ops._check(L.vx355_agg_get_output(h, out.descs, ncols, 4096, C.byref(n), C.byref(fin)))
        finally:
            L.vx355_agg_destroy(h)
        if not fin.value or n.value < 1:
            raise RuntimeError("config 1: %d groups, finished %d" % (n.value, fin.value))
        return n.value

    def rows_per_step(self):
        return self.n

    def info(self):
        if self.stream and hasattr(self, "submit_ms"):
            return {"input": "1000 x 10 000-row host vectors through vx355_agg_add_input_async (PCIe inclusive)",
                    "driver_thread_ms_queueing_the_last_step": round(self.submit_ms, 3),
                    "host_bytes_per_step": self.n * 16}
        return {"input": "1000 x 10 000-row host vectors (PCIe inclusive)" if self.stream
                else "one HBM-resident batch per step (five C-ABI calls per step: create, add_input, no_more_input, "
                     "get_output into host buffers, destroy), rotating through %d distinct 160 MB inputs (%d MB working "
                     "set: %s)" % (len(self.inputs), 160 * len(self.inputs),
                              "beyond the 256 MiB Infinity Cache, every step reads HBM" if len(self.inputs) >= 2
                              else "the replayed input sits in the Infinity Cache")}

    def host_sample(self, rows):
        rows = min(rows, self.n)
        out = {"k": self.k[:rows].cpu().numpy(), "v": self.v[:rows].cpu().numpy()}
        if hasattr(self, "v_valid"):
            out["v_valid"] = self.v_valid[:rows].cpu().numpy()
        return out

    def cpu_reference(self, sample, oracle):
        t0 = time.perf_counter()
        batch = abi.HostBatch([abi.HostColumn(abi.BIGINT, sample["k"]),
                               abi.HostColumn(abi.DOUBLE, sample["v"], sample.get("v_valid"))])
        op = oracle.Aggregation([0], [abi.BIGINT], self.AGGS)
        op.add_input(batch)
        op.no_more_input()
        out = oracle.collect_output(op, 4096)
        return out, time.perf_counter() - t0


# ---------------------------------------------------------------- C4 ----------
class C4(C1):
    """BASELINE configs[3]: 1 B rows, 100 M distinct BIGINT keys, sum(DOUBLE)."""
    name = "c4_groupby_1b_100m"
    bytes_per_row = 40
    agg_bytes_per_row = 40
    dominant = "k_agg_global"
    AGGS = [(abi.AGG_SUM, 1, abi.DOUBLE)]
    # Radix-partitioned LDS path (DESIGN.md "high cardinality"): algorithmic bytes per
    # input row of each pass, records are 16 bytes {key|row|mask, operand}.
    PASS_BYTES = {"k_rp_count1": 8, "k_rp_scatter1": 16 + 16, "k_rp_count2": 16, "k_rp_scatter2": 16 + 16,
                  "k_rp_aggregate": 16, "k_agg_global": 40}

    # sparse keys (open-addressing table): records carry the full key as a third word
    PASS_BYTES_SPARSE = {"k_rp_count1": 8, "k_rp_scatter1": 16 + 24, "k_rp_count2": 24, "k_rp_scatter2": 24 + 24,
                         "k_rp_aggregate": 24, "k_agg_global": 40}

    # no group order wanted, dense keys: 12-byte records {32-bit key | mask word, operand} (compact_record_launches)
    PASS_BYTES_COMPACT = {"k_rp_count1": 8, "k_rp_scatter1": 16 + 12, "k_rp_count2": 12, "k_rp_scatter2": 12 + 12,
                          "k_rp_aggregate": 12, "k_agg_global": 40}
    compact = False

    # ... sparse keys: 16-byte records {key, operand}, the home slot recomputed from the key
    PASS_BYTES_SPARSE_COMPACT = {"k_rp_count1": 8, "k_rp_scatter1": 16 + 16, "k_rp_count2": 16, "k_rp_scatter2": 16 + 16,
                                 "k_rp_aggregate": 16, "k_agg_global": 40}

    def pass_table(self):
        if getattr(self, "sparse", False):
            return self.PASS_BYTES_SPARSE_COMPACT if self.compact else self.PASS_BYTES_SPARSE
        return self.PASS_BYTES_COMPACT if self.compact else self.PASS_BYTES

    def pick_dominant(self, prof):
        """The slowest pass is the roofline kernel of this workload."""
        table = self.pass_table()
        name = max(table, key=lambda k: prof.get(k, (0.0, 0))[0])
        self.dominant = name
        self.agg_bytes_per_row = table[name]

    def __init__(self, torch, n, device, seed):
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        self.n = n
        distinct = int(os.environ.get("VX355_C4_DISTINCT", max(1, n // 10)))
        self.k = torch.randint(0, distinct, (n,), dtype=torch.int64, device=device, generator=g)
        if getattr(self, "sparse", False):
            # splitmix64 finaliser in wrapping int64 arithmetic (logical shifts by masking).
            z = self.k * -7046029254386353131  # 0x9E3779B97F4A7C15
            z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * -4658895280553007687  # 0xBF58476D1CE4E5B9
            z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * -7723592293110705685  # 0x94D049BB133111EB
            self.k = z ^ ((z >> 31) & ((1 << 33) - 1))
            del z
        self.v = torch.rand(n, dtype=torch.float64, device=device, generator=g)
        self.batch = DevBatch([dcol(abi.BIGINT, self.k), dcol(abi.DOUBLE, self.v)], n)
        torch.cuda.synchronize()

    unordered = False

    def info(self):
        return {"input": "one HBM-resident batch (16 GB)",
                "group_order": "not requested (VX355_AGG_UNORDERED_OUTPUT)" if self.unordered else "first-seen order"}

    def step(self, step_kind=abi.STEP_SINGLE):
        op = ops.HashAggregation([0], [abi.BIGINT], self.AGGS, step_kind,
                                 flags=abi.AGG_UNORDERED_OUTPUT if self.unordered else 0)
        op.add_input(self.batch)
        op.no_more_input()
        # 100 M groups: drain into HBM-resident output buffers, 16 M rows at a time.
        torch = __import__("torch")
        cap = 1 << 24
        if not hasattr(self, "_out"):
            dev = self.k.device
            self._out = (torch.empty(cap, dtype=torch.int64, device=dev),
                         torch.empty(cap, dtype=torch.float64, device=dev),
                         torch.empty(cap // 64 + 1, dtype=torch.int64, device=dev),
                         torch.empty(cap // 64 + 1, dtype=torch.int64, device=dev))
        descs = (abi.OutColumn * 2)()
        descs[0].type_kind, descs[0].mem = abi.BIGINT, abi.MEM_DEVICE
        descs[0].values, descs[0].nulls = self._out[0].data_ptr(), self._out[2].data_ptr()
        descs[1].type_kind, descs[1].mem = abi.DOUBLE, abi.MEM_DEVICE
        descs[1].values, descs[1].nulls = self._out[1].data_ptr(), self._out[3].data_ptr()
        total = 0
        while True:
            n, fin = C.c_int32(), C.c_int32()
            ops._check(ops.lib().vx355_agg_get_output(op.h, descs, 2, cap, C.byref(n), C.byref(fin)))
            total += n.value
            if fin.value:
                break
        self.groups = total
        self.compact = op.stats().compact_record_launches > 0
        return total

    def cpu_reference(self, sample, oracle):
        t0 = time.perf_counter()
        batch = abi.HostBatch([abi.HostColumn(abi.BIGINT, sample["k"]), abi.HostColumn(abi.DOUBLE, sample["v"])])
        op = oracle.Aggregation([0], [abi.BIGINT], self.AGGS)
        op.add_input(batch)
        op.no_more_input()
        n = 0
        while True:
            _, got, fin = op.get_output(1 << 20)
            n += got
            if fin:
                break
        return n, time.perf_counter() - t0


# ---------------------------------------------------------------- Q3 ----------
class Q3:
    """BASELINE configs[2], join part: orders (filtered) build, lineitem probe —
    the second, dominant join of TPC-H Q3 (TpchQueryBuilder.cpp:467-558) with
    TPC-H-shaped keys: 150 M orders with sparse keys (8 of every 32), o_orderdate
    < 1995-03-15 kept (~48.6 %), then ~20 % survive the customer semi-join ->
    ~14.6 M build rows; 600 M lineitems, l_shipdate > 1995-03-15 (~54 %) probe."""
    name = "tpch_q3_sf100_join"
    bytes_per_row = 24
    # What k_join_probe has to move per probe row in array mode: the 8-byte key in, the
    # 4-byte hit out (the presence bitmap and the head words of the ~1 % matching rows are
    # cache resident / negligible). SURVEY.md §8(d) prices the reference's layout instead:
    # key + one 16-byte table slot = 24 B/probe, reported next to it.
    agg_bytes_per_row = 12
    contract_bytes_per_row = 24
    dominant = "k_join_probe"
    Q3_DATE = 9204  # 1995-03-15
    random_probe = False

    def __init__(self, torch, n, device, seed):
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        self.torch, self.n = torch, n
        n_orders = max(8, n // 4)
        seq = torch.arange(n_orders, dtype=torch.int64, device=device)
        okey = (seq // 8) * 32 + (seq % 8)
        odate = torch.randint(8035, 10440, (n_orders,), dtype=torch.int32, device=device, generator=g)
        keep = (odate < self.Q3_DATE) & (torch.rand(n_orders, device=device, generator=g) < 0.2)
        self.bkey = okey[keep].contiguous()
        self.bdate = odate[keep].contiguous()
        self.bprio = torch.zeros_like(self.bdate)
        if self.random_probe:
            # worst case for the probe: lineitems in random order
            li = torch.randint(0, n_orders, (n,), dtype=torch.int64, device=device, generator=g)
        else:
            # dbgen order: lineitem is clustered by l_orderkey, 1..7 lines per order
            counts = torch.randint(1, 8, (n_orders,), dtype=torch.int64, device=device, generator=g)
            li = torch.repeat_interleave(seq, counts)
            del counts
        n = int(li.shape[0])
        self.n = n
        lkey = okey[li]
        lship = odate[li] + torch.randint(1, 122, (n,), dtype=torch.int32, device=device, generator=g)
        del li, okey, odate, seq
        sel = lship > self.Q3_DATE
        self.pkey = lkey[sel].contiguous()
        self.probe_rows = int(self.pkey.shape[0])
        del lkey, lship, sel
        self.build = DevBatch([dcol(abi.BIGINT, self.bkey), dcol(abi.INTEGER, self.bdate),
                               dcol(abi.INTEGER, self.bprio)], int(self.bkey.shape[0]))
        self.probe = DevBatch([dcol(abi.BIGINT, self.pkey)], self.probe_rows)
        cap = self.probe_rows
        self.mapping = torch.empty(cap, dtype=torch.int32, device=device)
        self.brows = torch.empty(cap, dtype=torch.int32, device=device)
        self.odate_out = torch.empty(cap, dtype=torch.int32, device=device)
        self.odate_nulls = torch.empty(cap // 64 + 1, dtype=torch.int64, device=device)
        torch.cuda.synchronize()

    def step(self, step_kind=None):
        b = ops.HashBuild([0], [abi.BIGINT], [1, 2], [abi.INTEGER, abi.INTEGER], abi.JOIN_INNER)
        b.add_input(self.build)
        table = b.finish()
        p = ops.HashProbe(table, [0], abi.JOIN_INNER)
        p.add_input(self.probe)
        descs = (abi.OutColumn * 1)()
        descs[0].type_kind, descs[0].mem = abi.INTEGER, abi.MEM_DEVICE
        descs[0].values, descs[0].nulls = self.odate_out.data_ptr(), self.odate_nulls.data_ptr()
        n, fin = p.get_output_device(self.probe_rows, self.mapping.data_ptr(), self.brows.data_ptr(),
                                     descs, [0])
        assert fin
        self.matches = n
        self.stats = table.stats()
        return n

    def pick_dominant(self, prof):
        """The roofline kernel of the step = the slowest kernel of the probe phase, priced at the bytes
        IT has to move per probe row: k_join_probe_list (probe pass that also lists the hits: 8-byte key
        in, 8 bytes out per MATCH), k_join_probe (4-byte hit out per probe row, k_emit lists later), or
        — range-partitioned probe for scattered keys — k_pp_count (8 in), k_pp_scatter (8 in, 8-byte
        record out), k_join_probe_part (8-byte record in, bitmap slice from LDS)."""
        per_row = {"k_join_probe_list": 8 + 8.0 * self.matches / max(1, self.probe_rows), "k_join_probe": 12,
                   "k_pp_count": 8, "k_pp_scatter": 16, "k_join_probe_part": 8 + 8.0 * self.matches / max(1, self.probe_rows)}
        name = max(per_row, key=lambda k: prof.get(k, (0.0, 0))[0])
        self.dominant = name
        self.agg_bytes_per_row = per_row[name]
        self.probe_phase_ms = sum(prof.get(k, (0.0, 0))[0] for k in per_row)

    def rows_per_step(self):
        return self.probe_rows

    def info(self):
        return {"build_rows": int(self.bkey.shape[0]), "probe_rows": self.probe_rows,
                "matches": int(self.matches), "table_mode": int(self.stats.hash_mode),
                "table_capacity": int(self.stats.capacity),
                "probe_order": "random" if self.random_probe else "dbgen (clustered by l_orderkey)"}

    def host_sample(self, rows):
        rows = min(rows, self.probe_rows)
        return {"bkey": self.bkey.cpu().numpy(), "bdate": self.bdate.cpu().numpy(),
                "pkey": self.pkey[:rows].cpu().numpy()}

    def cpu_reference(self, sample, oracle):
        """Build and probe timed SEPARATELY. The build side is the whole build side (a join's table does
        not shrink with the probe sample), the probe side a sample: the seconds returned are the probe
        sample's own time plus the sample's SHARE of the build time, so that sample rows / seconds is the
        rate of the whole step (build + probe of every row) - what the GPU figure divides by."""
        t0 = time.perf_counter()
        b = oracle.JoinBuild([0], [abi.BIGINT], [1], [abi.INTEGER], abi.JOIN_INNER)
        b.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, sample["bkey"]),
                                   abi.HostColumn(abi.INTEGER, sample["bdate"])]))
        t = b.finish()
        build_s = time.perf_counter() - t0
        t1 = time.perf_counter()
        p = oracle.JoinProbe(t, [0], abi.JOIN_INNER)
        total = 0
        pk = sample["pkey"]
        for lo in range(0, len(pk), 1 << 20):
            p.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, pk[lo:lo + (1 << 20)])]))
            while True:
                m, r, cols, fin = p.get_output(1 << 20)
                total += len(m)
                if fin:
                    break
        probe_s = time.perf_counter() - t1
        share = len(pk) / float(self.probe_rows)
        self.cpu_detail = {"build_rows": int(len(sample["bkey"])), "build_s": build_s,
                           "build_rows_per_s": len(sample["bkey"]) / build_s,
                           "probe_sample_rows": int(len(pk)), "probe_s": probe_s, "probe_rows_per_s": len(pk) / probe_s,
                           "value_is": "probe rows / (probe time + the sample's share of the build time): the rate of "
                                       "the whole step, like the GPU figure"}
        return total, probe_s + build_s * share

    def cpu_reference_mt(self, oracle, cores, sample_rows):
        """One thread per physical core, the reference's shape (exec/HashBuild.cpp:819-993,
        exec/HashTable.cpp:1003-1203): every build Driver fills its own row container from its slice of
        the build rows, the last one merges them with parallelJoinBuild (rows partitioned by bucket range,
        one inserter per partition); then every probe Driver probes its slice of the sample against the
        shared table."""
        from concurrent.futures import ThreadPoolExecutor
        cores = max(2, min(cores, 128))
        bkey, bdate = self.bkey.cpu().numpy(), self.bdate.cpu().numpy()
        per_probe = max(250_000, min(2_000_000, sample_rows // 4))
        rows = min(self.probe_rows, per_probe * cores)
        per_probe = rows // cores
        pkey = self.pkey[:rows].cpu().numpy()
        nb = len(bkey)
        cuts = [nb * i // cores for i in range(cores + 1)]

        def fill(i):
            b = oracle.JoinBuild([0], [abi.BIGINT], [1], [abi.INTEGER], abi.JOIN_INNER)
            b.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, bkey[cuts[i]:cuts[i + 1]]),
                                       abi.HostColumn(abi.INTEGER, bdate[cuts[i]:cuts[i + 1]])]))
            return b

        def probe(i, table):
            p = oracle.JoinProbe(table, [0], abi.JOIN_INNER)
            pk = pkey[i * per_probe:(i + 1) * per_probe]
            total = 0
            for lo in range(0, len(pk), 1 << 20):
                p.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, pk[lo:lo + (1 << 20)])]))
                while True:
                    m, r, cols, fin = p.get_output(1 << 20)
                    total += len(m)
                    if fin:
                        break
            return total
        oracle.set_join_build_threads(cores)
        try:
            with ThreadPoolExecutor(max_workers=cores) as ex:
                t0 = time.perf_counter()
                builds = list(ex.map(fill, range(cores)))
                table = builds[0].finish(builds[1:])
                build_s = time.perf_counter() - t0
                probe_s = None
                for _ in range(2):   # best of two: the first pass pays thread start-up and page faults
                    t1 = time.perf_counter()
                    list(ex.map(lambda i: probe(i, table), range(cores)))
                    d = time.perf_counter() - t1
                    probe_s = d if probe_s is None else min(probe_s, d)
        finally:
            oracle.set_join_build_threads(1)
        sample = per_probe * cores
        seconds = probe_s + build_s * (sample / float(self.probe_rows))
        return {"value": sample / seconds, "unit": "probe rows/s", "cores": cores, "kind": "port",
                "build_rows": nb, "build_s": build_s, "build_rows_per_s": nb / build_s,
                "probe_sample_rows": sample, "probe_s": probe_s, "probe_rows_per_s": sample / probe_s,
                "sample": f"{cores} build threads (own row containers, then HashTable::parallelJoinBuild restated: "
                          f"oracle/table.h) over all {nb} build rows, {cores} probe threads x {per_probe} probe rows "
                          "against the shared table; value = probe rows / (probe time + the sample's share of the "
                          "build time)",
                "host_cores_available": os.cpu_count()}


class Q3Full:
    """BASELINE configs[2], the whole query (TpchQueryBuilder.cpp:467-558) through
    velox_amd/tpch.py: customer (15 M) -> build; orders (150 M) -> probe -> build;
    lineitem (~600 M, dbgen order) -> probe -> 3-key aggregation. rows = customer +
    orders + lineitem rows scanned per step; scan bytes 24 + 24 + 28 B/row."""
    name = "tpch_q3_sf100_full_query"
    # both probes read their flat key column and the date column of the fused filter:
    # 8-byte key + 4-byte date in (+ 4-byte hit out for the dense form)
    agg_bytes_per_row = 16
    contract_bytes_per_row = 24
    dominant = "k_join_probe"
    fuse_filters = True

    def __init__(self, torch, n, device, seed):
        from velox_amd import tpch
        self.tpch, self.torch = tpch, torch
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        n_orders = max(8, n // 4)
        n_cust = max(3, n_orders // 10)
        t = {}
        t["c_custkey"] = torch.arange(1, n_cust + 1, dtype=torch.int64, device=device)
        seg_words = {  # 16-byte StringViews of the five market segments (size, 4-byte prefix, 8-byte tail)
            0: b"AUTOMOBILE", 1: b"BUILDING", 2: b"FURNITURE", 3: b"MACHINERY", 4: b"HOUSEHOLD"}
        views = np.zeros((5, 16), dtype=np.uint8)
        for i, sname in seg_words.items():
            views[i, 0:4] = np.frombuffer(np.uint32(len(sname)).tobytes(), dtype=np.uint8)
            views[i, 4:4 + len(sname)] = np.frombuffer(sname, dtype=np.uint8)
        seg_table = torch.from_numpy(views.view(np.int32).reshape(5, 4)).to(device)
        t["c_mktsegment"] = seg_table[torch.randint(0, 5, (n_cust,), device=device, generator=g)].contiguous()
        seq = torch.arange(n_orders, dtype=torch.int64, device=device)
        t["o_orderkey"] = (seq // 8) * 32 + (seq % 8)
        ck = torch.randint(0, (n_cust * 2) // 3, (n_orders,), dtype=torch.int64, device=device, generator=g)
        t["o_custkey"] = (ck // 2) * 3 + 1 + (ck % 2)          # the 2/3 of custkeys that are not 0 mod 3
        t["o_orderdate"] = torch.randint(8035, 10440, (n_orders,), dtype=torch.int32, device=device, generator=g)
        t["o_shippriority"] = torch.zeros(n_orders, dtype=torch.int32, device=device)
        counts = torch.randint(1, 8, (n_orders,), dtype=torch.int64, device=device, generator=g)
        li = torch.repeat_interleave(seq, counts)
        nl = int(li.shape[0])
        t["l_orderkey"] = t["o_orderkey"][li]
        t["l_shipdate"] = t["o_orderdate"][li] + torch.randint(1, 122, (nl,), dtype=torch.int32, device=device,
                                                                generator=g)
        del li, counts, seq, ck
        qty = torch.randint(1, 51, (nl,), dtype=torch.int32, device=device, generator=g).to(torch.float64)
        t["l_extendedprice"] = qty * (torch.randint(90000, 210001, (nl,), dtype=torch.int32, device=device,
                                                    generator=g).to(torch.float64) / 100.0)
        t["l_discount"] = torch.randint(0, 11, (nl,), dtype=torch.int32, device=device, generator=g).to(
            torch.float64) / 100.0
        del qty
        self.t = t
        self.rows = n_cust + n_orders + nl
        self.scan_bytes = 24 * n_cust + 24 * n_orders + 28 * nl
        torch.cuda.synchronize()

    @property
    def bytes_per_row(self):
        return self.scan_bytes / self.rows

    def step(self, step_kind=None):
        out, info = self.tpch.run_q3(ops, self.torch, self.t, fuse_filters=self.fuse_filters)
        self.last_info = info
        self.selected = info["orders_selected"] + info["lineitems_selected"]   # rows the two probes see
        return out

    def rows_per_step(self):
        return self.rows

    def pick_dominant(self, prof):
        """The two probes run under different profile names: orders -> customer table through the
        dense probe (k_join_probe), lineitem -> orders table through the listing probe
        (k_join_probe_list). The slower one is the roofline kernel, priced on ITS rows."""
        dense = prof.get("k_join_probe", (0.0, 0))[0]
        listing = prof.get("k_join_probe_list", (0.0, 0))[0]
        if listing >= dense:
            self.dominant = "k_join_probe_list"
            self.selected = self.last_info["lineitems_selected"]
            self.agg_bytes_per_row = 12   # 8-byte key + 4-byte l_shipdate (fused filter) in; only matches are written
        else:
            self.dominant = "k_join_probe"
            self.selected = self.last_info["orders_selected"]
            self.agg_bytes_per_row = 16   # 8-byte key + 4-byte o_orderdate in, 4-byte hit out

    def info(self):
        return dict(self.last_info, rows={k: int(v.shape[0]) for k, v in self.t.items()
                                          if k in ("c_custkey", "o_orderkey", "l_orderkey")})

    def host_sample(self, rows):
        # An SF-scaled copy of the same generator: rows/ (765 M / SF100) of the data.
        frac = max(1e-4, min(1.0, rows / float(self.rows)))
        small = Q3Full(self.torch, max(64, int((self.rows * frac) * 600 / 765)), self.t["c_custkey"].device, 4321)
        host = {k: v.cpu().numpy() for k, v in small.t.items()}
        host["_rows"] = small.rows
        return host

    def cpu_reference(self, sample, oracle):
        """numpy filters + oracle joins + oracle aggregation, one thread."""
        t0 = time.perf_counter()
        date = self.tpch.Q3_DATE
        seg = sample["c_mktsegment"].view(np.uint8).reshape(-1, 16)
        building = np.zeros(16, dtype=np.uint8)
        building[0] = 8
        building[4:12] = np.frombuffer(b"BUILDING", dtype=np.uint8)
        csel = np.flatnonzero((seg == building).all(axis=1))
        b1 = oracle.JoinBuild([0], [abi.BIGINT], [], [], abi.JOIN_INNER)
        b1.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, sample["c_custkey"][csel])]))
        t1 = b1.finish()
        osel = np.flatnonzero(sample["o_orderdate"] < date)
        p1 = oracle.JoinProbe(t1, [0], abi.JOIN_INNER)
        p1.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, sample["o_custkey"][osel])]))
        maps = []
        while True:
            m, r, cols, fin = p1.get_output(1 << 20, [])
            maps.append(m)
            if fin:
                break
        oidx = osel[np.concatenate(maps)] if maps else osel[:0]
        b2 = oracle.JoinBuild([0], [abi.BIGINT], [1, 2], [abi.INTEGER, abi.INTEGER], abi.JOIN_INNER)
        b2.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, sample["o_orderkey"][oidx]),
                                   abi.HostColumn(abi.INTEGER, sample["o_orderdate"][oidx]),
                                   abi.HostColumn(abi.INTEGER, sample["o_shippriority"][oidx])]))
        t2 = b2.finish()
        lsel = np.flatnonzero(sample["l_shipdate"] > date)
        p2 = oracle.JoinProbe(t2, [0], abi.JOIN_INNER)
        p2.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, sample["l_orderkey"][lsel])]))
        maps, od, op_ = [], [], []
        while True:
            m, r, cols, fin = p2.get_output(1 << 20)
            maps.append(m)
            od.append(cols[0][0])
            op_.append(cols[1][0])
            if fin:
                break
        lidx = lsel[np.concatenate(maps)]
        rev = sample["l_extendedprice"][lidx] * (1 - sample["l_discount"][lidx])
        agg = oracle.Aggregation([0, 1, 2], [abi.BIGINT, abi.INTEGER, abi.INTEGER], [(abi.AGG_SUM, 3, abi.DOUBLE)])
        agg.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, sample["l_orderkey"][lidx]),
                                     abi.HostColumn(abi.INTEGER, np.concatenate(od)),
                                     abi.HostColumn(abi.INTEGER, np.concatenate(op_)),
                                     abi.HostColumn(abi.DOUBLE, rev)]))
        agg.no_more_input()
        groups = 0
        while True:
            _, got, fin = agg.get_output(1 << 20)
            groups += got
            if fin:
                break
        return groups, time.perf_counter() - t0


class C5:
    """BASELINE configs[4]: fact (fk BIGINT, m DOUBLE) join dim (pk BIGINT unique, a BIGINT),
    both row-range partitioned over the GPUs; radix repartition by VectorHasher
    hash, one all-to-all per column over RCCL, local build + probe
    (velox_amd/dist.py: repartitioned_join). rows = fact rows per GPU; the dim
    side has rows / 10 per GPU. At N = 1 the exchange is a local copy."""
    name = "c5_partitioned_join"
    bytes_per_row = 32
    agg_bytes_per_row = 24
    dominant = "k_join_probe"

    def __init__(self, torch, n, device, seed, rank=0, world=1):
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        self.torch, self.n, self.world = torch, n, world
        nd = max(1, n // 10)
        self.pk = (torch.arange(rank * nd, (rank + 1) * nd, dtype=torch.int64, device=device) * 7919) % (1 << 45)
        self.a = torch.randint(0, 1 << 40, (nd,), dtype=torch.int64, device=device, generator=g)
        idx = torch.randint(0, world * nd, (n,), dtype=torch.int64, device=device, generator=g)
        self.fk = (idx * 7919) % (1 << 45)
        self.m = torch.rand(n, dtype=torch.float64, device=device, generator=g)
        # What the join must deliver for this rank's fact rows: sum (mod 2^64) of dim.a over the
        # partner of every fk. Every rank's `a` is the first draw of its seeded generator, so
        # any rank can restate all of them.
        expected = 0
        for r in range(world):
            if r == rank:
                a_r = self.a
            else:
                g_r = torch.Generator(device=device)
                g_r.manual_seed(seed - rank + r)
                a_r = torch.randint(0, 1 << 40, (nd,), dtype=torch.int64, device=device, generator=g_r)
            mine = (idx >= r * nd) & (idx < (r + 1) * nd)
            expected += int((a_r[(idx - r * nd).clamp_(0, nd - 1)] * mine).sum().item())
            del a_r, mine
        self.expected_payload_sum = expected & ((1 << 64) - 1)
        self.payload_sum = None   # set by a step run with self.checking
        self.checking = False
        del idx
        self.backend = vdist.GpuJoinBackend(ops, torch)
        torch.cuda.synchronize()

    comm = None    # ops.Comm: the in-library exchange (vx355_join_repartition)
    tdist = None   # else: torch.distributed (module or GroupDist) through velox_amd/dist.py

    def step(self, step_kind=None):
        if self.comm is not None:
            return self.step_library()
        dist = self.tdist
        if self.world > 1:
            # probe side in 4 chunks: the all-to-all of one chunk overlaps the partitioning of the next
            per_chunk, table = vdist.repartitioned_join_pipelined(self.backend, dist, self.torch,
                                                                  [self.pk, self.a], [self.fk, self.m], chunks=4)
            total = sum(int(m.shape[0]) for _, outs in per_chunk for m, _ in outs)
            stats = table.stats()
            if self.checking:
                self.payload_sum = sum(int(p.sum().item()) for _, outs in per_chunk for _, p in outs)
        else:
            fn = None
            if os.environ.get("VX355_C5_LIBEXCHANGE") == "1":   # torch-owned buffers, library collectives
                if not hasattr(self, "_libcomm"):
                    self._libcomm = ops.Comm(ops.Comm.unique_id(), 1, 0)
                fn = vdist.LibExchange(self.torch, self._libcomm)
            total, outputs, stats = vdist.repartitioned_join(self.backend, dist, self.torch,
                                                             [self.pk, self.a], [self.fk, self.m], exchange_fn=fn)
            if self.checking:
                self.payload_sum = sum(int(p.sum().item()) for _, p in outputs)
        self.matches, self.stats = total, stats
        return total

    def step_library(self):
        """The whole repartitioned join inside libvx355 (vx355_join_repartition): hash, partition,
        group by destination, exchange, build; probe side in 4 pipelined chunks. The sink drains
        every chunk's probe into HBM-resident output buffers (mapping, build rows, payload a)."""
        torch = self.torch
        # chunks pipeline the probe side against the links (chunk i + 1 travels while chunk i is probed);
        # one rank has no links to hide, and every chunk is one more pass over the join table
        chunks = int(os.environ.get("VX355_C5_CHUNKS", "4" if self.world > 1 else "1"))
        self.chunks = chunks
        cap = int(self.n // chunks * 1.25) + (1 << 20)
        if not hasattr(self, "_out"):
            dev = self.fk.device
            self._out = (torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev),
                         torch.empty(cap, dtype=torch.int64, device=dev), torch.empty(cap // 64 + 1, dtype=torch.int64, device=dev))
            self._build = DevBatch([dcol(abi.BIGINT, self.pk), dcol(abi.BIGINT, self.a)], int(self.pk.shape[0]))
            self._probe = DevBatch([dcol(abi.BIGINT, self.fk), dcol(abi.DOUBLE, self.m)], self.n)
        mapping, brows, payload, nulls = self._out
        descs = (abi.OutColumn * 1)()
        descs[0].type_kind, descs[0].mem = abi.BIGINT, abi.MEM_DEVICE
        descs[0].values, descs[0].nulls = payload.data_ptr(), nulls.data_ptr()
        total, sums = [0], []

        def sink(chunk, received, probe):
            while True:
                got, fin = probe.get_output_device(cap, mapping.data_ptr(), brows.data_ptr(), descs, [0])
                total[0] += got
                if self.checking:   # (get_output_device has synchronised the probe's stream)
                    sums.append(int(payload[:got].sum().item()))
                if fin:
                    break
        table = ops.join_repartition(self.comm, ([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER), self._build,
                                     ([0], abi.JOIN_INNER), self._probe, chunks, sink)
        self.matches, self.stats = total[0], table.stats()
        if self.checking:
            self.payload_sum = sum(sums)
        return total[0]

    def pick_dominant(self, prof):
        """The probe kernel of the step: k_join_probe_grouped when the chunks were regrouped by slice of
        the join table (vx355_join_probe_add_input_regrouped), else k_join_probe; SURVEY 8(d): 24 B per
        probe row (8-byte key read, one 16-byte slot, the hit)."""
        self.dominant = "k_join_probe_grouped" if prof.get("k_join_probe_grouped", (0.0, 0))[0] > 0 else "k_join_probe"

    def verify(self):
        """One untimed step with the outputs summed: (sum of the gathered dim.a, expected), mod 2^64."""
        self.checking = True
        try:
            self.step()
        finally:
            self.checking = False
        return self.payload_sum & ((1 << 64) - 1), self.expected_payload_sum

    def rows_per_step(self):
        return self.n

    def info(self):
        return {"fact_rows_per_gpu": self.n, "dim_rows_per_gpu": int(self.pk.shape[0]),
                "matches_on_rank0": int(self.matches), "table_mode": int(self.stats.hash_mode),
                "probe_chunks": getattr(self, "chunks", None)}

    def host_sample(self, rows):
        rows = min(rows, self.n)
        return {"pk": self.pk.cpu().numpy(), "a": self.a.cpu().numpy(), "pkey": self.fk[:rows].cpu().numpy()}

    def cpu_reference(self, sample, oracle):
        t0 = time.perf_counter()
        b = oracle.JoinBuild([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER)
        b.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, sample["pk"]), abi.HostColumn(abi.BIGINT, sample["a"])]))
        t = b.finish()
        p = oracle.JoinProbe(t, [0], abi.JOIN_INNER)
        total = 0
        pk = sample["pkey"]
        for lo in range(0, len(pk), 1 << 20):
            p.add_input(abi.HostBatch([abi.HostColumn(abi.BIGINT, pk[lo:lo + (1 << 20)])]))
            while True:
                m, r, cols, fin = p.get_output(1 << 20)
                total += len(m)
                if fin:
                    break
        return total, time.perf_counter() - t0


WORKLOADS = {"q1x4": (Q1FourKeys, 600_037_902), "q3full": (Q3Full, 600_037_902), "c5": (C5, 1_000_000_000), "q1": (Q1, 600_037_902), "c1": (C1, 10_000_000), "c4": (C4, 1_000_000_000),
             "q3": (Q3, 600_037_902)}


def measured_ceilings():
    """What this box's HBM delivers to plain streaming kernels, next to the 8 TB/s datasheet peak
    (SURVEY.md section 8(d)): the library's own read-only stream (the shape of k_agg_fast: bytes in,
    nothing out) and copy (bytes in, as many out: the scatter passes) kernels, vx355_hbm_ceiling.
    A kernel is compared with the ceiling of ITS shape."""
    return {"read_GBps": ops.hbm_ceiling(abi.CEILING_READ, 8 << 30, 5),
            "copy_GBps": ops.hbm_ceiling(abi.CEILING_COPY, 4 << 30, 5),
            # TPC-H Q1's scan with nothing behind it: seven column streams in k_agg_fast's row -> lane
            # mapping (68 B/row, 20 GB = the 300 M rows of one launch); interleaved streams deliver less
            # than one, and this - not the single stream - is what the kernel's arithmetic competes with
            "q1_columns_GBps": ops.hbm_ceiling(abi.CEILING_READ_COLUMNS, 68 * 300_000_000, 5),
            "how": "vx355_hbm_ceiling: 16-byte nontemporal accesses; read = 8 GiB read-only stream (four per lane in "
                   "flight, 8 workgroups of 512 per CU), copy = 4 GiB read + 4 GiB written (eight per lane in flight, "
                   "4 workgroups of 1024 per CU, one contiguous 2 MiB-aligned range each: tools/copy_bench.hip); "
                   "q1_columns = the seven column streams of TPC-H Q1's scan (68 B/row, 300 M rows) in k_agg_fast's "
                   "row -> lane mapping with nothing computed (tools/q1_stream_bench.hip)"}


READ_ONLY_KERNELS = ("k_agg_fast", "k_agg_lds", "k_join_probe", "k_join_probe_list", "k_join_probe_grouped", "k_rp_count1", "k_pp_count")


def pmc_traffic(workload, kernel):
    """Fallback for roofline.traffic when rocprofv3 is not available to bench.py: HBM bytes
    per STEP of the dominant kernel from the committed rocprofv3 PMC passes."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return {}
    try:
        entry = json.load(open(files[-1])).get(workload, {})
    except (OSError, ValueError):
        return {}
    return entry if entry.get("kernel") == kernel else {}


FETCH_CORRECTION = 2.0   # MI355X_MICROARCH.md "HBM": gfx950 FETCH_SIZE reports half of the bytes of wide reads


def measure_traffic(child_flags, kernel, steps=1):
    """HBM traffic of `kernel` per step, measured NOW: two child runs of this script under
    rocprofv3 (one --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE pass, each with --kernel-trace only,
    as MI355X_MICROARCH.md prescribes), `steps` timed steps and no warm-up each. FETCH_SIZE /
    WRITE_SIZE are KB per dispatch; summed over the kernel's dispatches, divided by the steps.
    Returns {} when rocprofv3 is missing or a pass fails (the caller falls back to profiles/)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    tool = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not tool:
        return {}
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="vx355_pmc_", dir="/tmp")
        cmd = [tool, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", tmp, "--",
               sys.executable, os.path.abspath(__file__)] + child_flags + [
                   "--steps", str(steps), "--warmup", "0", "--no-cpu-baseline", "--no-secondary", "--no-traffic",
                   "--counter-child"]
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                return {}
            kb, dispatches = 0.0, 0
            for path in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if row["Counter_Name"] == counter and any(k + "<" in row["Kernel_Name"] or k + "(" in row["Kernel_Name"]
                                                                  for k in kernel.split("|")):
                            kb += float(row["Counter_Value"])
                            dispatches += 1
            if dispatches == 0:
                return {}
            out[counter] = kb * 1024.0 / steps
            out["dispatches_per_step"] = dispatches / steps
        except (OSError, subprocess.SubprocessError, KeyError, ValueError):
            return {}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    fetch = out["FETCH_SIZE"] * FETCH_CORRECTION
    return {"traffic_bytes_per_step": fetch + out["WRITE_SIZE"], "fetch_bytes_per_step": fetch,
            "write_bytes_per_step": out["WRITE_SIZE"], "fetch_correction": FETCH_CORRECTION,
            "dispatches_per_step": out["dispatches_per_step"],
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child runs of this bench (measured in this run)"}


def cpu_model():
    """'model name' of /proc/cpuinfo (what lscpu prints): every CPU number carries it."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """Distinct (socket, core) pairs of /proc/cpuinfo; logical CPUs / 2 if unreadable."""
    try:
        pairs, phys = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                pairs.add((phys, line.split(":")[1].strip()))
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline_mt(wl, oracle_lib, sample_rows):
    """Joins: wl.cpu_reference_mt (parallel build + parallel probe). The reference's parallel shape for aggregations (one Driver per core, each
    with its own partial HashAggregation over a slice of the rows):
    one worker thread per physical core runs the single-thread CPU leg on its own
    slice (ctypes and numpy release the GIL). The final merge of the workers'
    groups is not included."""
    from concurrent.futures import ThreadPoolExecutor
    if hasattr(wl, "cpu_reference_mt"):
        block = wl.cpu_reference_mt(oracle_lib, min(physical_cores(), 128), sample_rows)
        block["cpu_model"] = cpu_model()
        return block
    cores = min(physical_cores(), 128)
    per = max(1_000_000, sample_rows // 8)
    sample = wl.host_sample(per * cores)
    total = len(next(v for k, v in sample.items() if not k.startswith("_")))
    cores = max(1, min(cores, total // per)) if total >= per else 1
    per = total // cores
    slices = [{k: (v[i * per:(i + 1) * per] if hasattr(v, "__len__") and len(v) == total else v)
               for k, v in sample.items()} for i in range(cores)]
    with ThreadPoolExecutor(max_workers=cores) as ex:
        dt = None
        for _ in range(3):  # best of three: the first pass pays thread start-up and page faults
            t0 = time.perf_counter()
            list(ex.map(lambda sl: wl.cpu_reference(sl, oracle_lib), slices))
            d = time.perf_counter() - t0
            dt = d if dt is None else min(dt, d)
    return {"value": per * cores / dt, "unit": "rows/s", "cores": cores, "kind": "port",
            "sample": f"{cores} worker threads x {per} rows of the same {wl.name} input, one partial "
                      "aggregation per worker (oracle/ restatement), best of 3 passes; final merge not included",
            "host_cores_available": os.cpu_count(), "cpu_model": cpu_model()}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch exactly one rank "
                         "per GPU (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N) or let "
                         "bench.py start the ranks itself (no WORLD_SIZE in the environment)")
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.launch_check:
        # CPU-only: the ranks exist, find each other and agree on the world size.
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([1], dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": int(t.item()), "requested_gpus": args.gpus}))
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # VX355_BENCH_SHARE_GPU=1 (debug on a 1-GPU box): every rank uses GPU 0 and the library's exchange
    # runs over its host shared-memory transport, because RCCL refuses two ranks per device.
    share = os.environ.get("VX355_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    if not share and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # Control plane (barriers, the max over ranks of the elapsed time, the communicator id) over
    # gloo; the DATA path is RCCL: libvx355's own communicator, or — when that is not usable —
    # a torch.distributed NCCL (= RCCL) group.
    backend = "gloo"
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ops.init(dev_index)
    comm, exchange_note = (None, "1 GPU")
    nccl_group = None
    if world > 1 or args.workload == "c5":
        comm, exchange_note = library_comm(args, torch, dist, rank, world, dev_index, share)
        if comm is None and args.exchange == "lib":
            raise SystemExit("--exchange lib: " + exchange_note)
        if comm is None and not share:
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_PORT", str(free_port()))
                dist.init_process_group("gloo", rank=rank, world_size=world)
            nccl_group = dist.new_group(backend="nccl", device_id=device)
            backend = "nccl"
    n_gpus = comm.info()[0] if comm is not None else (dist.get_world_size() if dist.is_initialized() else 1)
    if n_gpus != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the communicator spans {n_gpus} ranks")

    cls, default_rows = WORKLOADS[args.workload]
    n_total = args.rows or default_rows
    # weak: every GPU gets the per-GPU row count; strong: that row count is the WHOLE job
    n = n_total if (world == 1 or args.scaling == "weak") else max(1, n_total // world)
    if args.workload == "q3":
        cls.random_probe = args.q3_random_probe
    if args.workload == "q3full":
        cls.fuse_filters = not args.unfused
        if args.unfused:
            cls.name = "tpch_q3_sf100_full_query_unfused_filters"
    if args.workload == "c1":
        cls.stream = args.c1_stream
    if args.workload == "c4":
        cls.sparse = args.c4_sparse
        cls.unordered = args.c4_unordered
        if args.c4_sparse:
            cls.name = "c4_groupby_1b_100m_sparse_keys"
        if not args.c4_unordered:
            cls.name += "_first_seen_order"
    if args.workload == "c5":
        wl = cls(torch, n, device, seed=1234 + rank, rank=rank, world=world)
        wl.comm = comm
        wl.tdist = GroupDist(dist, nccl_group) if nccl_group is not None else dist
    else:
        wl = cls(torch, n, device, seed=1234 + rank)
    if args.workload == "q1":
        wl.fused = not args.unfused
        wl.stream = args.host_stream
        if args.host_stream:
            wl.name = "tpch_q1_streamed_host_vectors"

    if world > 1 and args.workload not in ("q1", "c5"):
        raise SystemExit(f"--workload {args.workload} is a single-GPU measurement; q1 and c5 shard over GPUs")

    def barrier():
        torch.cuda.synchronize()
        ops.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def q1_step_sharded(w):
        # N GPUs: every rank aggregates its own shard (Velox's partial step), the tiny partial
        # results meet on every rank and the final step merges them — the same partial / final
        # split the reference uses across Drivers and exchanges (docs/develop/aggregations.rst:24-91).
        if comm is not None:
            part = w.partial_operator()
            fin = ops.merge_partials(comm, part, list(range(len(Q1_KEYS[1]))), Q1_KEYS[1],
                                     vdist.final_aggs_for(Q1.FUSED_AGGS if w.fused else Q1_AGGS, len(Q1_KEYS[1])))
            return ops.collect_output(fin, 1024)
        part = w.step(abi.STEP_PARTIAL)
        group = GroupDist(dist, nccl_group) if nccl_group is not None else dist
        return vdist.merge_partials(ops, group, torch, part, Q1_KEYS[1], Q1.FUSED_AGGS if w.fused else Q1_AGGS,
                                    device if nccl_group is not None else None)

    def one_step():
        if world == 1 or args.workload != "q1":
            return wl.step()
        return q1_step_sharded(wl)

    def timed(step_fn, steps, warmup):
        for _ in range(warmup):
            step_fn()
        barrier()
        # EXACTLY 'steps' timed steps between two barriers, without the profiler's events (two per
        # launch: a third of config 1's step); the kernel times come from the same number of steps
        # repeated with the events on, outside the timed region
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        barrier()
        dt = time.perf_counter() - t0
        ops.profile_reset()
        if args.counter_child:
            return dt, {}
        ops.profile_enable(True)
        for _ in range(steps):
            step_fn()
        barrier()
        ops.profile_enable(False)
        prof_ = ops.profile()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, prof_

    elapsed, prof = timed(one_step, args.steps, args.warmup)
    rows = wl.rows_per_step() * world * args.steps
    result_check = None
    if hasattr(wl, "verify") and not args.counter_child:
        # the job's output against what the generator implies, over all ranks (not timed; not in a counter
        # child: its extra step would be counted into the traffic of the timed ones)
        got_sum, want_sum = wl.verify()
        if world > 1:
            both = [None] * world
            dist.all_gather_object(both, (got_sum, want_sum))
            got_sum = sum(b[0] for b in both) & ((1 << 64) - 1)
            want_sum = sum(b[1] for b in both) & ((1 << 64) - 1)
        result_check = {"what": "sum mod 2^64 of the joined dim.a over every fact row, all ranks",
                        "ok": got_sum == want_sum}
        if got_sum != want_sum:
            raise SystemExit(f"c5: the join's output checksum is {got_sum}, the generator implies {want_sum}")

    # N > 1, q1, weak scaling: the strong-scaling form of the same query (the N = 1 row count
    # sharded N ways = BASELINE's "SF100 at 1/2/4/8 GPUs") is measured next to the headline.
    strong = None
    if world > 1 and args.workload == "q1" and args.scaling == "weak" and not args.no_secondary:
        shard = max(1, n // world)
        wl_strong = wl.shard_view(shard)
        dt, _ = timed(lambda: q1_step_sharded(wl_strong), max(1, args.secondary_steps), 2)
        strong = {"scaling": "strong", "value": shard * world * max(1, args.secondary_steps) / dt, "unit": "rows/s",
                  "steps": max(1, args.secondary_steps), "warmup": 2,
                  "ms_per_step": dt / max(1, args.secondary_steps) * 1e3,
                  "rows_total": shard * world, "rows_per_gpu": shard}
    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    copy_ceiling = measured_ceilings()
    child_flags = ["--workload", args.workload]
    for flag, on in (("--q3-random-probe", args.q3_random_probe), ("--c4-sparse", args.c4_sparse),
                     ("--c4-unordered", args.c4_unordered),
                     ("--unfused", args.unfused), ("--c1-stream", args.c1_stream), ("--host-stream", args.host_stream)):
        if on:
            child_flags.append(flag)
    if args.rows:
        child_flags += ["--rows", str(args.rows)]
    measure = (not args.no_traffic) and world == 1
    out = {
        "metric": "rows/s + HBM GB/s (rocprof), TPC-H Q1 agg & Q3 join SF100, 1/2/4/8 MI355X",
        "value": rows / elapsed,
        "unit": "rows/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling if world > 1 else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": wl.name, "rows_per_gpu": wl.rows_per_step(),
                   "scan_bytes_per_row": wl.bytes_per_row,
                   "plan": ("fused FilterProject+HashAggregation (2 keys, 8 aggregates)" if wl.fused else
                            "FilterProject -> HashAggregation (2 keys, 8 aggregates)")
                   if args.workload == "q1" else (STEP_TEXT.get(args.workload, args.workload)),
                   "parallelism": ("one process per GPU, row shards; " +
                                   ("partial -> PrestoPages -> all-gather -> final" if args.workload == "q1" else
                                    "hash repartition of both sides, grouped send / recv per peer, local join") +
                                   " over RCCL") if world > 1 else "1 GPU",
                   "exchange": exchange_note},
        "workload_info": wl.info() if hasattr(wl, "info") else {},
        # N > 1: True = the in-library RCCL exchange was NOT used (its self-check failed or --exchange torch):
        # the scaling numbers of such a line measure torch.distributed's collectives, not the library's
        "exchange_downgraded": bool(world > 1 and comm is None),
        "result_check": result_check,
        "pipeline_algorithmic_GBps": wl.bytes_per_row * rows / elapsed / 1e9 / world,
        "roofline": roofline_block(wl, prof, args.steps, copy_ceiling, child_flags if measure else None),
        "kernels_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in sorted(prof.items())},
    }
    if strong is not None:
        out["strong_scaling"] = strong
    if not args.no_cpu_baseline and world == 1:   # the CPU legs are timed at N = 1 only
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        oracle_lib.lib()
        out["cpu_baseline"] = cpu_baseline_block(wl, oracle_lib, args.cpu_sample_rows, args.workload)
        if args.workload in ("q1", "q1x4", "c1", "c4", "q3") and not args.no_cpu_mt:
            out["cpu_baseline_mt"] = cpu_baseline_mt(wl, oracle_lib, args.cpu_sample_rows)
    if args.workload == "q1" and world == 1 and not args.no_secondary and not args.rows:
        # The rest of BASELINE's single-GPU configs next to the headline, each with its own roofline and
        # CPU baseline: Q1 in BASELINE.json's own wording (4 keys / 6 aggregates), config 1, the Q3 join
        # (the other half of the metric) in dbgen and in random probe order, the whole Q3, config 4 with
        # dense and with sparse keys. --secondary NAME[,NAME] restricts the list.
        del wl
        torch.cuda.empty_cache()
        out["secondary"] = {}
        wanted = set(args.secondary.split(",")) if args.secondary else None
        for key, workload, attrs, flags, steps, warmup in SECONDARY:
            if wanted is not None and key not in wanted:
                continue
            try:
                out["secondary"][key] = secondary_block(torch, device, args, copy_ceiling, measure, workload, attrs,
                                                        flags, steps, warmup)
            except Exception as e:   # a failing secondary must not take the headline with it; it is reported
                out["secondary"][key] = {"error": f"{type(e).__name__}: {e}"}
    C.CDLL(None).fflush(None)   # C stdio (RCCL prints a version banner there): the JSON line stays the last line
    emit(out, args.detail)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    C.CDLL(None).fflush(None)
    # Whatever a library prints while the process winds down goes to stderr: the JSON line stays the
    # last line of stdout. (Not os._exit: a profiler - rocprofv3 - writes its output at normal exit.)
    os.dup2(2, 1)


LINE_LIMIT = 4096   # the driver keeps an 8 KB tail of stdout: the final line must fit it with room to spare


def _num(x, digits=6):
    """floats to 'digits' significant digits: the line is for reading and parsing, the detail file keeps all"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def compact_roofline(r):
    if not r:
        return None
    algo = r.get("algorithmic_bytes_per_step")
    traffic = r.get("traffic")
    return {"bound": r.get("bound"), "kernel": r.get("kernel"), "achieved": _num(r.get("achieved")),
            "peak": r.get("peak"), "unit": r.get("unit"), "frac": _num(r.get("frac"), 4),
            "traffic": _num(traffic), "algorithmic_bytes": _num(algo),
            "traffic_ratio": _num(traffic / algo, 4) if (traffic and algo) else None,
            "kernel_ms_per_step": _num(r.get("kernel_ms_per_step"), 5),
            "launches_per_step": r.get("launches_per_step"),
            "measured_read_ceiling_GBps": _num((r.get("measured_ceiling") or {}).get("read_GBps"), 5),
            "measured_copy_ceiling_GBps": _num((r.get("measured_ceiling") or {}).get("copy_GBps"), 5),
            "measured_q1_columns_ceiling_GBps": _num((r.get("measured_ceiling") or {}).get("q1_columns_GBps"), 5),
            "frac_of_q1_columns_ceiling": _num(r.get("frac_of_q1_columns_ceiling"), 4),
            "ceiling_for_this_kernel": r.get("measured_ceiling_kind")}


def compact_cpu(c):
    if not c:
        return None
    return {"value": _num(c.get("value")), "unit": c.get("unit"), "cores": c.get("cores"), "kind": c.get("kind"),
            "sample": (c.get("sample") or "")[:160]}


def compact_line(out):
    """The final stdout line: the contract's keys, the headline's roofline and cpu_baseline, and every
    secondary block reduced to {value, ms_per_step, kernel, frac, traffic_ratio, cpu_value}. Everything
    else (per-kernel times, pass tables, notes) is in the detail object printed before it / written to
    bench_detail.json. Guaranteed shorter than LINE_LIMIT bytes: fields are dropped, never truncated."""
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                    "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["value"] = _num(line["value"], 8)
    line["ms_per_step"] = _num(line["ms_per_step"], 6)
    cfg = dict(out.get("config") or {})
    for k in ("plan", "parallelism", "exchange"):
        if isinstance(cfg.get(k), str):
            cfg[k] = cfg[k][:120]
    line["config"] = cfg
    line["roofline"] = compact_roofline(out.get("roofline"))
    line["cpu_baseline"] = compact_cpu(out.get("cpu_baseline"))
    if out.get("cpu_baseline_mt"):
        mt = out["cpu_baseline_mt"]
        line["cpu_baseline_mt"] = {"value": _num(mt.get("value")), "unit": mt.get("unit"), "cores": mt.get("cores"),
                                   "kind": mt.get("kind")}
    if out.get("exchange_downgraded"):
        line["exchange_downgraded"] = True
    if out.get("result_check") is not None:
        line["result_check_ok"] = bool(out["result_check"].get("ok"))
    if out.get("strong_scaling"):
        st = out["strong_scaling"]
        line["strong_scaling"] = {"value": _num(st.get("value"), 8), "ms_per_step": _num(st.get("ms_per_step")),
                                  "rows_total": st.get("rows_total")}
    if out.get("secondary"):
        sec = {}
        for key, blk in out["secondary"].items():
            if "error" in blk:
                sec[key] = {"error": str(blk["error"])[:80]}
                continue
            r = blk.get("roofline") or {}
            algo, traffic = r.get("algorithmic_bytes_per_step"), r.get("traffic")
            sec[key] = {"value": _num(blk.get("value")), "ms_per_step": _num(blk.get("ms_per_step"), 5),
                        "kernel": r.get("kernel"), "frac": _num(r.get("frac"), 4),
                        "traffic_ratio": _num(traffic / algo, 4) if (traffic and algo) else None,
                        "cpu_value": _num((blk.get("cpu_baseline") or {}).get("value"))}
            if blk.get("cpu_baseline_mt"):
                sec[key]["cpu_mt_value"] = _num(blk["cpu_baseline_mt"].get("value"))
                sec[key]["cpu_mt_cores"] = blk["cpu_baseline_mt"].get("cores")
            if blk.get("host_ingest"):
                sec[key]["host_GBps"] = _num(blk["host_ingest"].get("GBps"), 4)
        line["secondary"] = sec
    line["detail"] = out.get("detail_file")
    text = json.dumps(line, separators=(",", ":"))
    # shed optional fields until the line fits (never happens with today's blocks; the guard is the contract)
    for victim in ("secondary", "cpu_baseline_mt", "strong_scaling"):
        if len(text) < LINE_LIMIT:
            break
        if victim == "secondary" and "secondary" in line:
            line["secondary"] = {k: {"value": v.get("value"), "ms_per_step": v.get("ms_per_step"), "frac": v.get("frac")}
                                 for k, v in line["secondary"].items()}
            text = json.dumps(line, separators=(",", ":"))
            if len(text) < LINE_LIMIT:
                break
        line.pop(victim, None)
        text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:
        cfg = line["config"]
        line["config"] = {"workload": cfg.get("workload")}
        line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:60] if line.get("cpu_baseline") else None
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, len(text)
    return text


def emit(out, detail_path):
    """Full object -> detail file and stderr (tagged 'BENCH_DETAIL '); compact object -> the LAST stdout line."""
    if detail_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(out, f, indent=1)
            out["detail_file"] = detail_path
        except OSError as e:
            out["detail_file"] = None
            print(f"bench.py: could not write {detail_path}: {e}", file=sys.stderr)
    # the full object goes to stderr: stdout carries nothing after warm-up but the compact line
    print("BENCH_DETAIL " + json.dumps(out), file=sys.stderr, flush=True)
    print(compact_line(out), flush=True)


class GroupDist:
    """torch.distributed restricted to one process group (the NCCL = RCCL group of the data path
    when the default group is gloo), with the call shapes velox_amd/dist.py uses."""

    def __init__(self, dist, group):
        self.dist, self.group = dist, group

    def get_world_size(self):
        return self.dist.get_world_size(self.group)

    def all_gather(self, out, t):
        return self.dist.all_gather(out, t, group=self.group)

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None, async_op=False):
        return self.dist.all_to_all_single(out, inp, output_split_sizes, input_split_sizes, group=self.group,
                                           async_op=async_op)

    @property
    def ReduceOp(self):
        return self.dist.ReduceOp

    def all_reduce(self, t, op=None):
        return self.dist.all_reduce(t, op=op if op is not None else self.dist.ReduceOp.SUM, group=self.group)


def roofline_block(wl, prof, steps, copy_ceiling, child_flags):
    """roofline of the workload's dominant kernel. achieved = algorithmic bytes per step / the
    kernel's time per step (HIP events on the operator's own stream, vx355_profile_*); traffic =
    HBM bytes per STEP from rocprofv3 PMC passes (measured now when child_flags is given)."""
    if hasattr(wl, "pick_dominant"):
        wl.pick_dominant(prof)
    dom_ms, dom_launches = prof.get(wl.dominant, (0.0, 0))
    dom_rows = getattr(wl, "selected", wl.rows_per_step()) * steps
    achieved = (wl.agg_bytes_per_row * dom_rows / (dom_ms * 1e-3) / 1e9) if dom_ms > 0 else None
    symbol = {"k_join_probe_list": "k_join_probe", "k_join_probe_part": "k_pp_probe",
              "k_pp_scatter": "k_pp_scatter_fast|k_pp_scatter",
              "k_rp_scatter1": "k_rp_scatter1_sorted|k_rp_scatter1",
              "k_rp_scatter2": "k_rp_scatter2_opt|k_rp_scatter2_sorted|k_rp_scatter2",
              "k_rp_aggregate": "k_rp_aggregate_hashed|k_rp_aggregate"}.get(wl.dominant, wl.dominant)  # profile label -> kernel symbol
    pmc = measure_traffic(child_flags, symbol) if child_flags is not None else {}
    if not pmc:
        pmc = pmc_traffic(wl.name, wl.dominant)
    if achieved and achieved > HBM_PEAK_GBS:
        achieved = None   # the byte accounting does not describe this kernel: print nothing rather than > 1
    block = {
        "bound": "hbm", "kernel": wl.dominant,
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
        "traffic": pmc.get("traffic_bytes_per_step"),
        "traffic_basis": "HBM bytes per step (all launches of the kernel in one step); FETCH_SIZE x %g + WRITE_SIZE"
                         % FETCH_CORRECTION,
        "traffic_fetch": pmc.get("fetch_bytes_per_step"),
        "traffic_write": pmc.get("write_bytes_per_step"),
        "traffic_source": pmc.get("source"),
        "algorithmic_bytes_per_step": wl.agg_bytes_per_row * dom_rows / steps,
        "algorithmic_bytes_per_row": wl.agg_bytes_per_row,
        "measured_ceiling": copy_ceiling,
        "frac_of_measured_ceiling": ((achieved / copy_ceiling["read_GBps" if wl.dominant in READ_ONLY_KERNELS
                                                             else "copy_GBps"])
                                     if (achieved and copy_ceiling) else None),
        "measured_ceiling_kind": "read" if wl.dominant in READ_ONLY_KERNELS else "copy",
        "frac_of_q1_columns_ceiling": ((achieved / copy_ceiling["q1_columns_GBps"])
                                       if (achieved and copy_ceiling and copy_ceiling.get("q1_columns_GBps")
                                           and wl.name == "tpch_q1_sf100" and wl.dominant == "k_agg_fast")
                                       else None),
        "kernel_ms_per_step": dom_ms / steps,
        "avg_launch_ms": (dom_ms / dom_launches) if dom_launches else None,
        "launches_per_step": dom_launches / steps,
    }
    contract = getattr(wl, "contract_bytes_per_row", None)
    if contract and achieved:
        # SURVEY.md section 8(d) prices the reference's layout (key + 16-byte slot = 24 B/probe). The
        # same time priced at those bytes; omitted when it would exceed the peak (the design simply
        # does not move those bytes).
        at_contract = achieved * contract / wl.agg_bytes_per_row
        block["contract_bytes_per_row"] = contract
        if at_contract <= HBM_PEAK_GBS:
            block["achieved_at_contract_bytes"] = at_contract
            block["frac_at_contract_bytes"] = at_contract / HBM_PEAK_GBS
    return block


def cpu_baseline_block(wl, oracle_lib, cpu_sample_rows, workload):
    sample = wl.host_sample(cpu_sample_rows)
    if "_rows" in sample:
        sample_rows = int(sample["_rows"])
    else:
        sample_rows = len(sample["pkey"]) if "pkey" in sample else len(next(iter(sample.values())))
    wl.cpu_detail = None
    cpu_out, cpu_s = wl.cpu_reference(sample, oracle_lib)
    block = {
        "value": sample_rows / cpu_s, "unit": "rows/s", "cores": 1, "kind": "port",
        "sample": f"first {sample_rows} rows of the same {wl.name} input, single thread: "
                  "Velox-algorithm CPU restatement (oracle/) of the same plan"
                  + (" with numpy FilterProject" if workload == "q1" else "")
                  + (" (local join only: no exchange on the CPU side)" if workload == "c5" else "")
                  + ("; the whole build side is built (timed separately) and charged to the probe sample by its share"
                     if getattr(wl, "cpu_detail", None) else ""),
        "host_cores_available": os.cpu_count(), "cpu_model": cpu_model(),
    }
    if getattr(wl, "cpu_detail", None):
        block["join"] = wl.cpu_detail
    return block


SECONDARY = [
    # key in `secondary`, --workload, class attributes, child flags of the traffic passes, steps, warm-up
    ("tpch_q1_sf100_4key_6agg", "q1x4", {}, [], None, 2),
    ("c1_groupby_10m_1k", "c1", {"stream": False}, [], 50, 5),
    ("tpch_q3_sf100_join", "q3", {"random_probe": False}, [], None, 2),
    ("tpch_q3_sf100_join_random_probe_order", "q3", {"random_probe": True}, ["--q3-random-probe"], None, 2),
    ("tpch_q3_sf100_full_query", "q3full", {}, [], None, 2),
    # config 4: result sets are compared as unordered multisets, so the headline
    # variant asks for no group order (VX355_AGG_UNORDERED_OUTPUT: what a FINAL step / exchange / ORDER BY
    # above the operator needs); the first-seen-order form (+ one sort of the groups) is the extra
    ("c4_groupby_1b_100m", "c4", {"sparse": False, "unordered": True, "name": "c4_groupby_1b_100m"},
     ["--c4-unordered"], 3, 1),
    ("c4_groupby_1b_100m_sparse_keys", "c4", {"sparse": True, "unordered": True,
                                              "name": "c4_groupby_1b_100m_sparse_keys"},
     ["--c4-sparse", "--c4-unordered"], 3, 1),
    ("c4_groupby_1b_100m_first_seen_order", "c4", {"sparse": False, "unordered": False, "_no_traffic": True,
                                                   "name": "c4_groupby_1b_100m_first_seen_order"}, [], 3, 1),
    ("c4_groupby_1b_100m_sparse_keys_first_seen_order", "c4",
     {"sparse": True, "unordered": False, "_no_traffic": True,
      "name": "c4_groupby_1b_100m_sparse_keys_first_seen_order"}, ["--c4-sparse"], 3, 1),
    # the boundary north_star names: host RowVectors of 10 000 rows (PCIe inclusive; roofline = the link, not HBM)
    ("c1_groupby_10m_1k_streamed_host_vectors", "c1", {"stream": True}, None, 10, 2),
    ("tpch_q1_60m_rows_streamed_host_vectors", "q1", {"stream": True, "rows_override": 60_000_000,
                                                      "name": "tpch_q1_streamed_host_vectors"}, None, 3, 1),
]


def secondary_block(torch, device, args, copy_ceiling, measure, workload, attrs, child_flags, steps, warmup):
    """One entry of `secondary`: a BASELINE config other than the headline, timed like the headline
    (inputs resident, synchronised on both sides), with its own roofline (counter traffic from
    rocprofv3 child passes of the same workload) and CPU baseline."""
    cls, rows = WORKLOADS[workload]
    attrs = dict(attrs)
    if attrs.pop("_no_traffic", False):
        measure = False   # an extra variant: timed and profiled, no rocprofv3 counter passes of its own
    saved = {k: getattr(cls, k, None) for k in attrs}
    for k, v in attrs.items():
        setattr(cls, k, v)
    rows = attrs.get("rows_override", rows)
    wl = cls(torch, rows, device, seed=1234)
    host_stream = child_flags is None   # PCIe-inclusive blocks: no HBM roofline, no counter passes
    child_flags = child_flags or []
    if workload in ("q1", "q1x4"):
        wl.fused = True
    steps = max(1, steps or args.secondary_steps)
    for _ in range(warmup):
        wl.step()
    torch.cuda.synchronize()
    ops.synchronize()
    # the step time is taken WITHOUT the per-launch events of the profiler (two events around every
    # launch cost ~10 us each way: a third of config 1's whole step), the kernel times in a second,
    # profiled pass over the same steps
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize()
    ops.synchronize()
    elapsed = time.perf_counter() - t0
    ops.profile_reset()
    ops.profile_enable(True)
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize()
    ops.synchronize()
    ops.profile_enable(False)
    prof = ops.profile()
    child = ["--workload", workload] + child_flags
    block = {
        "value": wl.rows_per_step() * steps / elapsed, "unit": "probe rows/s" if workload == "q3" else "rows/s",
        "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "config": {"workload": wl.name, "rows": wl.rows_per_step(), "step": STEP_TEXT[workload]},
        "workload_info": wl.info() if hasattr(wl, "info") else {},
        "kernels_ms_per_step": {k: round(v[0] / steps, 4) for k, v in sorted(prof.items())},
    }
    if host_stream:
        nbytes = block["workload_info"].get("host_bytes_per_step", 0)
        block["host_ingest"] = {"GBps": nbytes * steps / elapsed / 1e9, "pcie_gen5_x16_GBps": 63.0,
                                "note": "PCIe inclusive: every vector starts in pageable host memory; not comparable "
                                        "with the HBM-resident lines"}
    else:
        block["roofline"] = roofline_block(wl, prof, steps, copy_ceiling, child if measure else None)
    if getattr(wl, "probe_phase_ms", None):
        # all probe-side kernels together, priced at key in + hit out (12 B/probe) and at SURVEY section 8(d)'s 24 B
        ms = wl.probe_phase_ms / steps
        block["probe_phase"] = {"ms_per_step": ms, "GBps_at_12B_per_probe": wl.rows_per_step() * 12 / (ms * 1e-3) / 1e9,
                                "GBps_at_24B_per_probe": wl.rows_per_step() * 24 / (ms * 1e-3) / 1e9}
    if workload == "c4":
        # every pass of the radix path next to the bytes it has to move (the step's roofline kernel is the slowest)
        table = wl.pass_table()
        block["record_bytes"] = (16 if wl.compact else 24) if getattr(wl, "sparse", False) else (12 if wl.compact else 16)
        block["passes"] = {k: {"ms_per_step": round(prof[k][0] / steps, 4), "algorithmic_bytes_per_row": table[k],
                               "algorithmic_GBps": table[k] * wl.rows_per_step() * steps / (prof[k][0] * 1e-3) / 1e9}
                           for k in table if prof.get(k, (0, 0))[0] > 0}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        oracle_lib.lib()
        block["cpu_baseline"] = cpu_baseline_block(wl, oracle_lib, min(args.cpu_sample_rows, 8_000_000), workload)
        if hasattr(wl, "cpu_reference_mt") and not args.no_cpu_mt:
            block["cpu_baseline_mt"] = cpu_baseline_mt(wl, oracle_lib, min(args.cpu_sample_rows, 8_000_000))
    del wl
    for k, v in saved.items():
        if v is None and k in cls.__dict__:
            delattr(cls, k)
        elif v is not None:
            setattr(cls, k, v)
    torch.cuda.empty_cache()
    return block


STEP_TEXT = {
    "q1": "fused FilterProject + HashAggregation, 2 keys / 8 aggregates (the reference's TPC-H Q1 plan)",
    "q1x4": "fused FilterProject + HashAggregation, 4 keys / 6 aggregates (BASELINE.json's wording of configs[1])",
    "c1": "HashAggregation k -> sum(v), count(*), one HBM-resident batch (BASELINE configs[0])",
    "q3": "HashBuild (add_input + finish) + HashProbe (add_input + get_output with one payload column), inner join",
    "q3full": "the whole TPC-H Q3: customer -> build; orders -> filter, probe, build; lineitem -> filter, probe; "
              "3-key aggregation (BASELINE configs[2])",
    "c4": "HashAggregation k -> sum(v) over 10^9 rows / 10^8 groups, groups drained into HBM pages (BASELINE configs[3]); "
          "group order not requested unless the name says first_seen_order",
}


if __name__ == "__main__":
    main()
