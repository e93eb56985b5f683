"""Multi-GPU (one process per GPU) plumbing for the sharded aggregation path.

The path shards by rows: every rank runs HashAggregation with step = PARTIAL
over its own shard; the partial results (one row per group, tiny for TPC-H Q1)
are all-gathered and merged by a step = FINAL operator on every rank — Velox's
own partial/final split (docs/develop/aggregations.rst:24-91), with
torch.distributed (RCCL on GPUs, gloo in the CPU tests) as the exchange. No
input row ever crosses a link.
"""
import numpy as np

from . import abi

MAX_GROUPS = 4096


def encode_partial(columns, kinds):
    """collect_output() columns -> float64 matrix [MAX_GROUPS, 2 * ncols + 1]
    (value, validity per column; last column marks live rows). Strings of up to
    7 bytes travel as their stringAsNumber image, BIGINT counts as doubles
    (exact below 2^53)."""
    ncols = len(columns)
    rows = len(columns[0][1]) if ncols else 0
    if rows > MAX_GROUPS:
        raise ValueError(f"{rows} partial groups exceed the gather buffer ({MAX_GROUPS})")
    out = np.zeros((MAX_GROUPS, 2 * ncols + 1))
    out[:rows, 2 * ncols] = 1
    for c, ((vals, valid), kind) in enumerate(zip(columns, kinds)):
        if kind in (abi.VARCHAR, abi.VARBINARY):
            enc = np.zeros(rows)
            for i, v in enumerate(vals):
                if v is not None:
                    if len(v) > 6:
                        raise ValueError("string keys longer than 6 bytes do not fit the gather encoding")
                    enc[i] = int.from_bytes(v, "little") + ((1 << (8 * len(v))) if len(v) else 0)
            vals = enc
        out[:rows, 2 * c] = np.asarray(vals, dtype=np.float64)
        out[:rows, 2 * c + 1] = np.asarray(valid, dtype=np.float64)
    return out


def decode_partials(matrix, kinds):
    """Gathered matrices (stacked) -> HostBatch of all ranks' partial rows."""
    ncols = len(kinds)
    live = matrix[matrix[:, 2 * ncols] == 1]
    cols = []
    for c, kind in enumerate(kinds):
        vals, valid = live[:, 2 * c], live[:, 2 * c + 1] > 0
        if kind in (abi.VARCHAR, abi.VARBINARY):
            strs = []
            for v in vals.astype(np.int64):
                v = int(v)
                if v == 0:
                    strs.append(b"")
                else:
                    size = (v.bit_length() - 1) // 8
                    strs.append((v - (1 << (8 * size))).to_bytes(size, "little"))
            cols.append(abi.HostColumn(kind, strs, valid))
        elif kind in abi.KIND_DTYPE:
            cols.append(abi.HostColumn(kind, vals.astype(abi.KIND_DTYPE[kind]), valid))
        else:
            raise ValueError(f"kind {kind} not supported by the gather encoding")
    return abi.HostBatch(cols, len(live))


def all_gather_partials(dist, torch, columns, kinds, device=None):
    """Every rank contributes its partial result; returns the HostBatch of all
    partial rows in rank order. device: a cuda device for RCCL, None for gloo."""
    mat = encode_partial(columns, kinds)
    t = torch.from_numpy(mat)
    if device is not None:
        t = t.to(device)
    gathered = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, t)
    allm = torch.cat(gathered).cpu().numpy()
    return decode_partials(allm, kinds)


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
    batch = all_gather_partials(dist, torch, partial_columns, kinds, device)
    op = impl.Aggregation(list(range(len(key_types))), list(key_types),
                          final_aggs_for(raw_aggs, len(key_types)), abi.STEP_FINAL)
    op.add_input(batch)
    op.no_more_input()
    return impl.collect_output(op, MAX_GROUPS)
