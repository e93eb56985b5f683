"""Multi-GPU (one process per GPU) plumbing for the sharded aggregation path.

The path shards by rows: every rank runs HashAggregation with step = PARTIAL
over its own shard; the partial results (one row per group, tiny for TPC-H Q1)
are all-gathered and merged by a step = FINAL operator on every rank — Velox's
own partial/final split (docs/develop/aggregations.rst:24-91), with
torch.distributed (RCCL on GPUs, gloo in the CPU tests) as the exchange. No
input row ever crosses a link.
"""
import os

import numpy as np

from . import abi

MAX_GROUPS = 4096


def encode_partial(columns, kinds, pad_rows=MAX_GROUPS):
    """collect_output() columns -> int64 matrix [pad_rows, 3 * ncols + 1]: two value
    words and a validity word per column, last column marks live rows. Every value
    travels as its own bits: BIGINT / INTEGER / ... as int64, DOUBLE / REAL bit-cast
    (float64.view(int64)), strings of up to 12 bytes (what the library keeps inline) as
    {size : 8 | bytes 0..6} and {bytes 7..11}. Nothing is rounded on the way: the
    FINAL step sees exactly what the PARTIAL step produced (sum(BIGINT) beyond 2^53,
    INT64 min / max, counts)."""
    ncols = len(columns)
    rows = len(columns[0][1]) if ncols else 0
    if rows > MAX_GROUPS or rows > pad_rows:
        raise ValueError(f"{rows} partial groups exceed the gather buffer ({min(pad_rows, MAX_GROUPS)})")
    out = np.zeros((pad_rows, 3 * ncols + 1), dtype=np.int64)
    out[:rows, 3 * ncols] = 1
    for c, ((vals, valid), kind) in enumerate(zip(columns, kinds)):
        valid = np.asarray(valid, dtype=bool)
        if kind in (abi.VARCHAR, abi.VARBINARY):
            for i, v in enumerate(vals):
                if v is None or not valid[i]:
                    continue
                v = bytes(v)
                if len(v) > 12:
                    raise ValueError("string keys longer than 12 bytes are not inline: not supported on device")
                w0 = len(v) | (int.from_bytes(v[:7], "little") << 8)
                out[i, 3 * c] = w0 - (1 << 64) if w0 >= (1 << 63) else w0
                out[i, 3 * c + 1] = int.from_bytes(v[7:], "little")
        elif kind in (abi.DOUBLE, abi.REAL):
            out[:rows, 3 * c] = np.asarray(vals, dtype=np.float64).view(np.int64)
        elif kind == abi.BOOLEAN or kind in abi.KIND_DTYPE:
            out[:rows, 3 * c] = np.asarray(vals).astype(np.int64)
        else:
            raise ValueError(f"kind {kind} not supported by the gather encoding")
        out[:rows, 3 * c + 2] = valid
    return out


def decode_partials(matrix, kinds):
    """Gathered matrices (stacked) -> HostBatch of all ranks' partial rows."""
    ncols = len(kinds)
    matrix = np.asarray(matrix, dtype=np.int64)
    live = matrix[matrix[:, 3 * ncols] == 1]
    cols = []
    for c, kind in enumerate(kinds):
        w0, w1, valid = live[:, 3 * c], live[:, 3 * c + 1], live[:, 3 * c + 2] != 0
        if kind in (abi.VARCHAR, abi.VARBINARY):
            strs = []
            for a, b in zip(w0.tolist(), w1.tolist()):
                a &= (1 << 64) - 1
                size = a & 0xFF
                raw = (a >> 8).to_bytes(7, "little") + (b & ((1 << 40) - 1)).to_bytes(5, "little")
                strs.append(raw[:size])
            cols.append(abi.HostColumn(kind, strs, valid))
        elif kind in (abi.DOUBLE, abi.REAL):
            vals = np.ascontiguousarray(w0).view(np.float64)
            cols.append(abi.HostColumn(kind, vals.astype(abi.KIND_DTYPE[kind]), valid))
        elif kind == abi.BOOLEAN:
            cols.append(abi.HostColumn(kind, w0 != 0, valid))
        elif kind in abi.KIND_DTYPE:
            cols.append(abi.HostColumn(kind, w0.astype(abi.KIND_DTYPE[kind]), valid))
        else:
            raise ValueError(f"kind {kind} not supported by the gather encoding")
    return abi.HostBatch(cols, len(live))


SMALL_GROUPS = 64


def _inline_only(columns, kinds):
    """True when the fixed-width matrix can carry every value (strings of <= 12 bytes)."""
    for (vals, valid), kind in zip(columns, kinds):
        if kind in (abi.VARCHAR, abi.VARBINARY):
            if any(v is not None and ok and len(v) > 12 for v, ok in zip(vals, valid)):
                return False
        elif kind != abi.BOOLEAN and kind not in abi.KIND_DTYPE:
            return False
    return True


def columns_to_batch(columns, kinds):
    """collect_output() / presto_deserialize() columns -> HostBatch."""
    cols = []
    rows = len(columns[0][1]) if columns else 0
    for (vals, valid), kind in zip(columns, kinds):
        valid = np.asarray(valid, dtype=bool)
        if kind in (abi.VARCHAR, abi.VARBINARY):
            vals = [bytes(v) if (v is not None and ok) else b"" for v, ok in zip(vals, valid)]
        elif kind == abi.TIMESTAMP:
            vals = np.asarray(vals, dtype=np.int64).reshape(-1, 2)
        else:
            vals = np.asarray(vals)
        cols.append(abi.HostColumn(kind, vals, valid))
    return abi.HostBatch(cols, rows)


def all_gather_partial_pages(impl, dist, torch, columns, kinds, device=None):
    """The partial rows of every rank as PrestoPages, the wire format Velox's own exchange moves
    (impl.presto_serialize / presto_deserialize: vx355_presto_serialize and _deserialize on the
    GPUs, the oracle's writer and an independent reader in the CPU tests): any number of groups,
    strings of any length, checksummed. Two collectives: page sizes, then the padded bytes."""
    rows = len(columns[0][1]) if columns else 0
    pages = impl.presto_serialize(columns_to_batch(columns, kinds), [0, rows], flags=abi.PAGE_CHECKSUM)
    page = pages[0] if pages else b""
    world = dist.get_world_size()

    def gather(t):
        if device is not None:
            t = t.to(device)
        got = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        return [g.cpu() for g in got]

    sizes = [int(t.item()) for t in gather(torch.tensor([len(page)], dtype=torch.int64))]
    padded = np.zeros(max(max(sizes), 1), dtype=np.uint8)
    padded[: len(page)] = np.frombuffer(page, dtype=np.uint8)
    received = gather(torch.from_numpy(padded))
    all_pages = [received[r].numpy()[: sizes[r]].tobytes() for r in range(world) if sizes[r]]
    _, cols = impl.presto_deserialize(all_pages, kinds)
    return columns_to_batch(cols, kinds)


def all_gather_partials(dist, torch, columns, kinds, device=None, impl=None):
    """Every rank contributes its partial result; returns the HostBatch of all
    partial rows in rank order. device: a cuda device for RCCL, None for gloo.

    One collective in the common case: a fixed [1 + SMALL_GROUPS, width] matrix per rank
    whose first row carries the rank's true group count. Only when some rank has more
    groups (or values the matrix cannot carry: strings beyond 12 bytes) does a second
    exchange follow: PrestoPages when impl offers the serializer, else a matrix sized by
    the largest count."""
    rows = len(columns[0][1]) if columns else 0
    width = 3 * len(kinds) + 1
    pages_ok = impl is not None and hasattr(impl, "presto_serialize") and hasattr(impl, "presto_deserialize")
    inline = _inline_only(columns, kinds)
    if not pages_ok:
        if rows > MAX_GROUPS:
            raise ValueError(f"{rows} partial groups exceed the gather buffer ({MAX_GROUPS})")
        if not inline:
            raise ValueError("string keys longer than 12 bytes need the page transport (impl with presto_serialize)")

    def gather(mat):
        t = torch.from_numpy(mat)
        if device is not None:
            t = t.to(device)
        gathered = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, t)
        return torch.stack(gathered).cpu().numpy()

    first = np.zeros((1 + SMALL_GROUPS, max(width, 2)), dtype=np.int64)
    first[0, 0] = rows
    first[0, 1] = 0 if inline else 1   # this rank needs the page transport
    if rows <= SMALL_GROUPS and inline:
        first[1:, :width] = encode_partial(columns, kinds, SMALL_GROUPS)
    got = gather(first)
    largest = int(got[:, 0, 0].max())
    if largest <= SMALL_GROUPS and not got[:, 0, 1].any():
        return decode_partials(got[:, 1:, :width].reshape(-1, width), kinds)
    if pages_ok:
        return all_gather_partial_pages(impl, dist, torch, columns, kinds, device)
    got = gather(encode_partial(columns, kinds, largest))
    return decode_partials(got.reshape(-1, width), kinds)


def final_aggs_for(raw_aggs, num_keys):
    """Aggregates of the FINAL step over the PARTIAL step's output layout:
    keys, then per aggregate one column (two for avg: sum, count)."""
    fin, col = [], num_keys
    for a in raw_aggs:
        kind, typ = a[0], a[2]
        if kind == abi.AGG_AVG:
            fin.append((abi.AGG_AVG, col, typ, -1, col + 1))
            col += 2
        else:
            fin.append((kind, col, typ))
            col += 1
    return fin


def partial_kinds(key_types, raw_aggs):
    kinds = list(key_types)
    for a in raw_aggs:
        kind, typ = a[0], a[2]
        if kind == abi.AGG_AVG:
            kinds += [abi.DOUBLE, abi.BIGINT]
        elif kind in (abi.AGG_COUNT, abi.AGG_COUNT_STAR):
            kinds.append(abi.BIGINT)
        elif kind == abi.AGG_SUM:
            kinds.append(abi.BIGINT if typ <= abi.BIGINT else abi.DOUBLE)
        else:
            kinds.append(typ)
    return kinds


def merge_partials(impl, dist, torch, partial_columns, key_types, raw_aggs, device=None):
    """All-gather + FINAL step. impl: module with Aggregation / collect_output
    (velox_amd.ops on GPUs)."""
    kinds = partial_kinds(key_types, raw_aggs)
    batch = all_gather_partials(dist, torch, partial_columns, kinds, device, impl)
    op = impl.Aggregation(list(range(len(key_types))), list(key_types),
                          final_aggs_for(raw_aggs, len(key_types)), abi.STEP_FINAL)
    op.add_input(batch)
    op.no_more_input()
    return impl.collect_output(op, max(MAX_GROUPS, batch.num_rows))


# ---- repartitioned hash join (BASELINE config 5) ------------------------------
# Both inputs start row-range partitioned over the ranks. Each rank hashes its
# key column (VectorHasher::hash), maps hashes to a destination rank
# (HashPartitionFunction::partition), groups its rows by destination
# (vx355_partition_scatter) and sends every group straight to its owner: ONE
# all-to-all per column (RCCL grouped send/recv over the direct xGMI links; gloo
# in the CPU tests). After the exchange equal keys live on the same rank and the
# join is local. Nothing else in the path communicates.

def partition_spec(world, top_bits=None):
    """(kind, kwargs) of the HashPartitionFunction flavour used for 'world'
    destinations: hash % world (exec/HashPartitionFunction.cpp:112-115), what a
    PartitionedOutput without a HashBitRange computes - a CPU Velox peer sends a row to
    the same rank. top_bits (default: VX355_EXCHANGE_TOP_BITS=1, as in libvx355's edge):
    the HashBitRange flavour for powers of two, the top log2(world) bits."""
    if top_bits is None:
        top_bits = os.environ.get("VX355_EXCHANGE_TOP_BITS", "0") not in ("", "0")
    if top_bits and world > 1 and world & (world - 1) == 0:
        bits = world.bit_length() - 1
        return abi.PART_BIT_RANGE, dict(bit_begin=64 - bits, bit_end=64)
    return abi.PART_MODULO, dict(num_partitions=max(1, world))


def _drain(torch, cols):
    """libvx355 runs on its own HIP streams and takes device buffers as complete
    (include/vx355.h, vx355_column.mem): after a collective or a torch kernel produced
    them, the producing stream must have finished before the library is called.
    dist.all_to_all_single / Work.wait() only order torch's CURRENT stream behind the
    communication stream; they do not block the host."""
    if cols and getattr(cols[0], "is_cuda", False):
        torch.cuda.current_stream(cols[0].device).synchronize()


MAX_MESSAGE_BYTES = 256 << 20


def _row_bytes(t):
    return t.element_size() * (t.shape[1] if t.dim() == 2 else 1)


def _all_to_all_rows(dist, torch, got, src, rc, sc):
    """dist.all_to_all_single(got, src, rc, sc) with every rank-to-rank message kept under
    MAX_MESSAGE_BYTES: the RCCL of this image (2.26.6) delivers only the first half of a message
    above 2^30 bytes - tools/torch_a2a_repro.py, profiles/r03_rccl_large_message.txt. One rank:
    a copy, no collective. Large slices travel as row-range pieces; every rank runs the same
    number of rounds (the largest slice anywhere decides), empty pieces included."""
    world = dist.get_world_size()
    if world == 1:
        got.copy_(src)
        return
    piece = max(1, MAX_MESSAGE_BYTES // _row_bytes(src))
    largest = torch.tensor([max(list(rc) + list(sc) + [0])], dtype=torch.int64, device=src.device)
    dist.all_reduce(largest, op=dist.ReduceOp.MAX)
    rounds = max(1, -(-int(largest.item()) // piece))
    if rounds == 1:
        dist.all_to_all_single(got, src, output_split_sizes=list(rc), input_split_sizes=list(sc))
        return
    s_off = [sum(sc[:i]) for i in range(world)]
    r_off = [sum(rc[:i]) for i in range(world)]
    for k in range(rounds):
        lo = k * piece
        sk = [max(0, min(piece, n - lo)) for n in sc]
        rk = [max(0, min(piece, n - lo)) for n in rc]
        part = torch.cat([src.narrow(0, s_off[i] + lo, sk[i]) for i in range(world) if sk[i] > 0] or [src[:0]])
        tmp = torch.empty((sum(rk),) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_to_all_single(tmp, part, output_split_sizes=rk, input_split_sizes=sk)
        at = 0
        for i in range(world):
            if rk[i] > 0:
                got.narrow(0, r_off[i] + lo, rk[i]).copy_(tmp.narrow(0, at, rk[i]))
                at += rk[i]


def exchange(dist, torch, cols, counts):
    """cols: torch tensors whose rows are grouped by destination rank;
    counts[r] = rows for rank r. Returns the received columns (source-rank
    order) and the per-source row counts."""
    world = dist.get_world_size()
    dev = cols[0].device if cols else "cpu"
    send = torch.tensor([int(c) for c in counts], dtype=torch.int64, device=dev)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send)
    rc = [int(x) for x in recv.cpu().tolist()]
    sc = [int(x) for x in counts]
    assert len(sc) == world
    out = []
    for c in cols:
        got = torch.empty((sum(rc),) + tuple(c.shape[1:]), dtype=c.dtype, device=c.device)
        _all_to_all_rows(dist, torch, got, c.contiguous(), rc, sc)
        out.append(got)
    _drain(torch, out)
    return out, rc


class LibExchange:
    """exchange() through libvx355's own RCCL communicator (vx355_exchange_counts /
    vx355_exchange_columns, include/vx355.h): the form a C++ host uses; torch only allocates
    the receive buffers here. comm: velox_amd.ops.Comm."""

    def __init__(self, torch, comm):
        self.torch, self.comm = torch, comm

    def __call__(self, cols, counts):
        torch = self.torch
        rc = self.comm.exchange_counts(counts)
        out = [torch.empty((sum(rc),) + tuple(c.shape[1:]), dtype=c.dtype, device=c.device) for c in cols]
        src = [c.contiguous() for c in cols]
        _drain(torch, src + out)   # the library's stream does not order itself behind torch's
        widths = [c.element_size() * (c.shape[1] if c.dim() == 2 else 1) for c in src]
        self.comm.exchange_columns([c.data_ptr() for c in src], widths, counts, rc, [o.data_ptr() for o in out])
        return out, rc


def exchange_async(dist, torch, cols, counts):
    """exchange() split in two: the row counts travel first (one tiny blocking all-to-all),
    the column payloads are posted with async_op=True. Returns a function that waits for
    them and hands back (received columns, per-source counts) — between the two calls the
    caller is free to run device work for the next chunk."""
    world = dist.get_world_size()
    dev = cols[0].device if cols else "cpu"
    send = torch.tensor([int(c) for c in counts], dtype=torch.int64, device=dev)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send)
    rc = [int(x) for x in recv.cpu().tolist()]
    sc = [int(x) for x in counts]
    assert len(sc) == world
    out, works, keep = [], [], []
    # every rank must take the same branch: the largest slice anywhere decides
    widest = max([_row_bytes(c) for c in cols] + [1])
    flag = torch.tensor([max(rc + sc + [0]) * widest], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    big = int(flag.item()) > MAX_MESSAGE_BYTES
    for c in cols:
        src = c.contiguous()
        got = torch.empty((sum(rc),) + tuple(c.shape[1:]), dtype=c.dtype, device=c.device)
        if world == 1 or big:
            _all_to_all_rows(dist, torch, got, src, rc, sc)   # large slices: in pieces, not overlapped
        else:
            works.append(dist.all_to_all_single(got, src, output_split_sizes=rc, input_split_sizes=sc, async_op=True))
        out.append(got)
        keep.append(src)

    def wait():
        for w in works:
            w.wait()
        _drain(torch, out)
        keep.clear()
        return out, rc
    return wait


def repartitioned_join_pipelined(backend, dist, torch, build_cols, probe_cols, chunks=4):
    """Same result as repartitioned_join, with the probe side cut into 'chunks' row
    ranges so that the exchange of chunk i rides the xGMI links while the GPU hashes,
    partitions and scatters chunk i + 1 and probes chunk i - 1 (the
    exchange is ~60 % of the critical path of config 5 if it is not overlapped).
    backend additionally supplies build(build_cols) -> table and
    probe(table, probe_cols) -> outputs; returns ([(received probe columns, outputs)] per
    chunk, table): probe outputs refer to rows of the chunk they came from."""
    world = dist.get_world_size()
    parts = backend.partitions(build_cols[0], world)
    grouped, counts = backend.scatter(parts, world, build_cols)
    received, _ = exchange(dist, torch, grouped, counts)
    table = backend.build(received)
    n = int(probe_cols[0].shape[0])
    chunks = max(1, min(chunks, n)) if n else 1
    bounds = [(n * i // chunks) for i in range(chunks + 1)]
    results, pending = [], None
    for i in range(chunks):
        cols = [c[bounds[i]:bounds[i + 1]] for c in probe_cols]
        parts = backend.partitions(cols[0], world)
        grouped, counts = backend.scatter(parts, world, cols)
        wait = exchange_async(dist, torch, grouped, counts)      # chunk i is in flight from here on
        if pending is not None:
            got, _ = pending()
            results.append((got, backend.probe(table, got)))
        pending = wait
    got, _ = pending()
    results.append((got, backend.probe(table, got)))
    return results, table


def repartitioned_join(backend, dist, torch, build_cols, probe_cols, exchange_fn=None):
    """build_cols / probe_cols: lists of torch tensors, column 0 is the BIGINT
    join key. backend supplies the device work:
      backend.partitions(key_tensor, world) -> uint32 partition number per row
      backend.scatter(partitions, world, cols) -> (cols grouped by partition, counts)
      backend.join(build_cols, probe_cols) -> whatever the caller wants back
    """
    world = dist.get_world_size()
    sides = []
    for cols in (build_cols, probe_cols):
        parts = backend.partitions(cols[0], world)
        grouped, counts = backend.scatter(parts, world, cols)
        received, _ = exchange_fn(grouped, counts) if exchange_fn else exchange(dist, torch, grouped, counts)
        sides.append(received)
    return backend.join(sides[0], sides[1])


class GpuJoinBackend:
    """Device work of repartitioned_join on the MI355X through libvx355."""

    def __init__(self, ops_module, torch, join_type=abi.JOIN_INNER):
        self.ops, self.torch, self.join_type = ops_module, torch, join_type

    def _dcol(self, kind, t):
        return self.ops.DeviceColumn.from_ptr(kind, t.data_ptr(), t.shape[0])

    def _batch(self, cols, kinds):
        import ctypes as C
        descs = (abi.Column * len(cols))(*[self._dcol(k, t).descriptor() for k, t in zip(kinds, cols)])
        batch = abi.Batch(int(cols[0].shape[0]), len(cols), descs)

        class B:
            def ref(self_inner):
                return C.byref(batch)
        b = B()
        b._keep = (descs, batch, cols)
        return b

    def partitions(self, key, world):
        torch, ops = self.torch, self.ops
        n = int(key.shape[0])
        hashes = torch.empty(n, dtype=torch.int64, device=key.device)
        parts = torch.empty(n, dtype=torch.int32, device=key.device)
        b = self._batch([key], [abi.BIGINT])
        ops._check(ops.lib().vx355_hash_columns(b.ref(), abi.i32_array([0]), 1, None, 0,
                                                hashes.data_ptr(), abi.MEM_DEVICE))
        kind, kw = partition_spec(world)
        ops._check(ops.lib().vx355_partition(hashes.data_ptr(), n, kind, kw.get("num_partitions", 0),
                                             kw.get("bit_begin", 0), kw.get("bit_end", 0),
                                             parts.data_ptr(), abi.MEM_DEVICE))
        return parts

    def scatter(self, parts, world, cols):
        torch, ops = self.torch, self.ops
        outs = [torch.empty_like(c) for c in cols]
        widths = [c.element_size() * (c.shape[1] if c.dim() == 2 else 1) for c in cols]
        counts = ops.partition_scatter_device(parts.data_ptr(), int(parts.shape[0]), world,
                                              [c.data_ptr() for c in cols], widths,
                                              [o.data_ptr() for o in outs])
        return outs, counts

    def build(self, build_cols):
        torch, ops = self.torch, self.ops
        kinds = {torch.int64: abi.BIGINT, torch.int32: abi.INTEGER, torch.float64: abi.DOUBLE}
        bkinds = [kinds[c.dtype] for c in build_cols]
        deps = list(range(1, len(build_cols)))
        build = ops.HashBuild([0], [abi.BIGINT], deps, [bkinds[i] for i in deps], self.join_type)
        build.add_input(self._batch(build_cols, bkinds))
        table = build.finish()
        table.payload_dtype = build_cols[1].dtype if deps else None
        table.payload_kind = bkinds[1] if deps else None
        return table

    def join(self, build_cols, probe_cols):
        """Inner join; returns (matches, [(mapping, gathered first payload)], table stats)."""
        table = self.build(build_cols)
        outputs = self.probe(table, probe_cols)
        return sum(int(m.shape[0]) for m, _ in outputs), outputs, table.stats()

    def probe(self, table, probe_cols):
        """-> [(mapping tensor, gathered first payload tensor or None)] for one probe batch."""
        torch, ops = self.torch, self.ops
        deps = [0] if table.payload_kind is not None else []
        probe = ops.HashProbe(table, [0], self.join_type)
        probe.add_input(self._batch([probe_cols[0]], [abi.BIGINT]))
        cap = max(1, int(probe_cols[0].shape[0]))
        dev = probe_cols[0].device
        mapping = torch.empty(cap, dtype=torch.int32, device=dev)
        rows = torch.empty(cap, dtype=torch.int32, device=dev)
        payload = torch.empty(cap, dtype=table.payload_dtype, device=dev) if deps else None
        nulls = torch.empty(cap // 64 + 1, dtype=torch.int64, device=dev)
        descs = None
        if deps:
            descs = (abi.OutColumn * 1)()
            descs[0].type_kind, descs[0].mem = table.payload_kind, abi.MEM_DEVICE
            descs[0].values, descs[0].nulls = payload.data_ptr(), nulls.data_ptr()
        outputs = []
        while True:
            n, fin = probe.get_output_device(cap, mapping.data_ptr(), rows.data_ptr(), descs,
                                             [0] if deps else [])
            outputs.append((mapping[:n].clone(), payload[:n].clone() if deps else None))
            # the clones run on torch's stream; the next get_output_device overwrites
            # mapping / payload on the probe's own stream
            _drain(torch, [mapping])
            if fin:
                break
        return outputs
