"""Python host side over the C ABI of libvx355 (include/vx355.h).

Mirrors the reference's operator interface for the hot path — HashAggregation
(exec/HashAggregation.h), HashBuild / HashProbe (exec/HashBuild.h,
exec/HashProbe.h): addInput / noMoreInput / getOutput — plus the standalone
VectorHasher / filter-compaction / partition kernels. Every call goes through
ctypes into the HIP library; there is no CPU fallback: loading fails loudly
when libvx355.so is missing and init() fails when no GPU is visible.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VX355_LIB_PATH") or os.path.join(_HERE, "libvx355.so")
_LIB = None

# Every symbol include/vx355.h declares.
SYMBOLS = [
    "vx355_init", "vx355_shutdown", "vx355_abi_version", "vx355_device_count", "vx355_last_error",
    "vx355_device_malloc", "vx355_device_free", "vx355_memcpy_h2d", "vx355_memcpy_d2h",
    "vx355_memset_d", "vx355_synchronize", "vx355_profile_enable", "vx355_profile_reset",
    "vx355_profile_get", "vx355_profile_names", "vx355_hash_columns", "vx355_value_ids",
    "vx355_filter_compact", "vx355_partition", "vx355_partition_scatter", "vx355_presto_serialize", "vx355_presto_deserialize", "vx355_filter_project", "vx355_agg_create", "vx355_agg_set_fused_input", "vx355_agg_add_input",
    "vx355_agg_no_more_input", "vx355_agg_output_types", "vx355_agg_get_output",
    "vx355_agg_get_stats", "vx355_agg_destroy", "vx355_agg_add_input_async", "vx355_agg_poll", "vx355_agg_wait",
    "vx355_join_build_add_input_async", "vx355_join_build_poll", "vx355_join_build_wait", "vx355_join_build_create",
    "vx355_join_build_add_input", "vx355_join_build_finish", "vx355_join_build_destroy",
    "vx355_join_table_retain", "vx355_join_table_release", "vx355_join_table_get_stats",
    "vx355_join_probe_create", "vx355_join_probe_add_input", "vx355_join_probe_get_output",
    "vx355_join_probe_destroy", "vx355_join_probe_get_build_side_output", "vx355_join_table_key_filter", "vx355_join_table_key_filter_values",
    "vx355_bloom_num_blocks", "vx355_join_table_key_filter_bloom", "vx355_bloom_test",
    "vx355_set_device", "vx355_current_device", "vx355_stream_wait_event", "vx355_default_stream",
    "vx355_agg_stream", "vx355_join_build_stream", "vx355_join_probe_stream",
    "vx355_join_probe_set_filter", "vx355_join_probe_set_output_batch_bytes",
    "vx355_comm_get_unique_id", "vx355_comm_create", "vx355_comm_create_all", "vx355_comm_info",
    "vx355_comm_stream", "vx355_comm_destroy", "vx355_exchange_counts", "vx355_exchange_columns",
    "vx355_all_gather", "vx355_agg_flush", "vx355_agg_to_intermediate",
    "vx355_value_dict_create", "vx355_value_dict_compute", "vx355_value_dict_lookup", "vx355_value_dict_size",
    "vx355_value_dict_destroy",
    "vx355_all_gather_v", "vx355_exchange_create", "vx355_exchange_send", "vx355_exchange_receive",
    "vx355_exchange_stream", "vx355_exchange_destroy", "vx355_exchange_destinations", "vx355_join_repartition", "vx355_agg_merge_partials",
    "vx355_hbm_ceiling", "vx355_compose_indices", "vx355_agg_table_bytes", "vx355_join_probe_set_input_filter",
    "vx355_join_probe_add_input_async", "vx355_join_probe_poll", "vx355_join_probe_wait",
    "vx355_agg_no_more_input_async", "vx355_agg_get_output_async", "vx355_agg_output_result",
    "vx355_join_probe_get_output_async", "vx355_join_probe_output_result",
    "vx355_join_probe_add_input_regrouped",
    "vx355_agg_bytes_in_use", "vx355_agg_get_gpu_stats", "vx355_join_build_get_gpu_stats", "vx355_join_probe_get_gpu_stats",
    "vx355_presto_compress_page", "vx355_presto_uncompress_page",
    "vx355_set_memory_limit", "vx355_memory_usage",
]

# void (*vx355_output_done_fn)(void* arg, int status, int32_t num_rows, int32_t finished)
OUTPUT_DONE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int32, C.c_int32)

# int (*vx355_join_chunk_sink)(void* arg, int32_t chunk, const vx355_batch* received, vx355_join_probe* probe)
JOIN_CHUNK_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(abi.Batch), C.c_void_p)


class Vx355Error(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"vx355 status {status}: {msg}")
        self.status = status


def lib():
    """Loads libvx355.so. Raises if the HIP extension has not been built: the
    product path never substitutes a CPU implementation."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -m velox_amd.build or __graft_entry__.build())")
    L = C.CDLL(LIB_PATH)
    i32, i64, u64, vp, sz = C.c_int32, C.c_int64, C.c_uint64, C.c_void_p, C.c_size_t
    P = C.POINTER
    L.vx355_init.argtypes = [C.c_int]
    L.vx355_last_error.restype = C.c_char_p
    L.vx355_device_malloc.restype = vp
    L.vx355_device_malloc.argtypes = [sz]
    L.vx355_device_free.argtypes = [vp]
    L.vx355_memcpy_h2d.argtypes = [vp, vp, sz]
    L.vx355_memcpy_d2h.argtypes = [vp, vp, sz]
    L.vx355_memset_d.argtypes = [vp, C.c_int, sz]
    L.vx355_profile_enable.argtypes = [C.c_int]
    L.vx355_profile_get.argtypes = [C.c_char_p, P(C.c_double), P(i64)]
    L.vx355_profile_names.argtypes = [C.c_char_p, sz]
    L.vx355_hash_columns.argtypes = [P(abi.Batch), P(i32), i32, vp, i32, vp, i32]
    L.vx355_value_ids.argtypes = [P(abi.Batch), P(i32), P(abi.ValueIdSpec), i32, vp, i32, vp, vp,
                                  P(i32), i32]
    L.vx355_filter_compact.argtypes = [vp, vp, vp, i32, vp, P(i32), i32]
    L.vx355_partition.argtypes = [vp, i32, i32, i32, i32, i32, vp, i32]
    L.vx355_partition_scatter.argtypes = [vp, i32, i32, P(vp), P(i32), i32, P(vp), P(i64), i32]
    L.vx355_presto_serialize.argtypes = [P(abi.Batch), vp, i32, vp, i32, i32, vp, i64, i32, vp]
    L.vx355_presto_deserialize.argtypes = [vp, vp, i32, vp, i32, i32, vp, i64, vp, i64, vp]
    L.vx355_set_memory_limit.argtypes = [i64]
    L.vx355_memory_usage.argtypes = [P(i64), P(i64), P(i64)]
    L.vx355_presto_compress_page.argtypes = [vp, i64, i32, C.c_float, vp, i64, P(i64)]
    L.vx355_presto_uncompress_page.argtypes = [vp, i64, i32, vp, i64, P(i64)]
    L.vx355_filter_project.argtypes = [P(abi.Batch), P(abi.FilterTerm), i32, P(abi.Projection), i32,
                                       vp, P(i32), P(vp), P(vp), i32]
    L.vx355_agg_create.argtypes = [P(abi.AggSpec), P(vp)]
    L.vx355_agg_set_fused_input.argtypes = [vp, P(abi.FilterTerm), i32, P(abi.Projection), i32]
    L.vx355_agg_add_input.argtypes = [vp, P(abi.Batch)]
    for name in ("vx355_agg", "vx355_join_build", "vx355_join_probe"):
        getattr(L, name + "_add_input_async").argtypes = [vp, P(abi.Batch), P(i64)]
        getattr(L, name + "_poll").argtypes = [vp, P(i64), P(i64)]
        getattr(L, name + "_wait").argtypes = [vp]
    L.vx355_agg_no_more_input.argtypes = [vp]
    L.vx355_agg_output_types.argtypes = [vp, P(i32), i32, P(i32)]
    L.vx355_agg_get_output.argtypes = [vp, P(abi.OutColumn), i32, i32, P(i32), P(i32)]
    L.vx355_agg_get_stats.argtypes = [vp, P(abi.AggStats)]
    L.vx355_agg_table_bytes.argtypes = [vp, P(C.c_int64), P(C.c_int64)]
    L.vx355_agg_no_more_input_async.argtypes = [vp, P(C.c_int64)]
    L.vx355_agg_get_output_async.argtypes = [vp, P(abi.OutColumn), i32, i32, OUTPUT_DONE_FN, vp, P(C.c_int64)]
    L.vx355_agg_output_result.argtypes = [vp, C.c_int64, P(i32), P(i32)]
    L.vx355_agg_flush.argtypes = [vp]
    L.vx355_agg_to_intermediate.argtypes = [vp, P(abi.Batch), P(abi.OutColumn), i32]
    L.vx355_agg_destroy.argtypes = [vp]
    L.vx355_agg_destroy.restype = None
    L.vx355_join_build_create.argtypes = [P(abi.JoinBuildSpec), P(vp)]
    L.vx355_join_build_add_input.argtypes = [vp, P(abi.Batch)]
    L.vx355_join_build_finish.argtypes = [vp, P(vp), i32, P(vp)]
    L.vx355_join_build_destroy.argtypes = [vp]
    L.vx355_join_build_destroy.restype = None
    L.vx355_join_table_retain.argtypes = [vp]
    L.vx355_join_table_retain.restype = None
    L.vx355_join_table_release.argtypes = [vp]
    L.vx355_join_table_release.restype = None
    L.vx355_join_table_get_stats.argtypes = [vp, P(abi.JoinTableStats)]
    L.vx355_join_probe_create.argtypes = [vp, P(abi.JoinProbeSpec), P(vp)]
    L.vx355_join_probe_add_input.argtypes = [vp, P(abi.Batch)]
    L.vx355_join_probe_add_input_regrouped.argtypes = [vp, P(abi.Batch), P(vp), P(i32)]
    L.vx355_join_probe_get_output.argtypes = [vp, i32, vp, vp, i32, P(abi.OutColumn), P(i32), i32,
                                              P(i32), P(i32)]
    L.vx355_join_probe_get_build_side_output.argtypes = [vp, i32, vp, i32, P(abi.OutColumn), P(i32), i32,
                                                         P(i32), P(i32)]
    L.vx355_join_probe_get_output_async.argtypes = [vp, i32, i32, vp, vp, i32, P(abi.OutColumn), P(i32), i32,
                                                    OUTPUT_DONE_FN, vp, P(C.c_int64)]
    L.vx355_join_probe_output_result.argtypes = [vp, C.c_int64, P(i32), P(i32)]
    L.vx355_join_probe_destroy.argtypes = [vp]
    L.vx355_join_probe_destroy.restype = None
    L.vx355_join_table_key_filter.argtypes = [vp, i32, P(abi.KeyFilter)]
    L.vx355_join_table_key_filter_values.argtypes = [vp, i32, vp, i64, i32, P(i64)]
    L.vx355_bloom_num_blocks.restype = i64
    L.vx355_bloom_num_blocks.argtypes = [i64, C.c_double, i32]
    L.vx355_join_table_key_filter_bloom.argtypes = [vp, i32, i32, vp, i64, i32]
    L.vx355_bloom_test.argtypes = [vp, i64, i32, P(abi.Column), i32, vp, vp, i32]
    L.vx355_join_probe_set_filter.argtypes = [vp, P(abi.JoinFilterTerm), i32]
    L.vx355_join_probe_set_output_batch_bytes.argtypes = [vp, i64]
    L.vx355_join_probe_set_input_filter.argtypes = [vp, P(abi.FilterTerm), i32]
    L.vx355_comm_get_unique_id.argtypes = [vp]
    L.vx355_comm_create.argtypes = [vp, i32, i32, P(vp)]
    L.vx355_comm_create_all.argtypes = [i32, P(i32), P(vp)]
    L.vx355_comm_info.argtypes = [vp, P(i32), P(i32), P(i32)]
    L.vx355_comm_stream.restype = vp
    L.vx355_comm_stream.argtypes = [vp]
    L.vx355_comm_destroy.argtypes = [vp]
    L.vx355_comm_destroy.restype = None
    L.vx355_exchange_counts.argtypes = [vp, P(i64), P(i64)]
    L.vx355_exchange_columns.argtypes = [vp, P(vp), P(i32), i32, P(i64), P(i64), P(vp)]
    L.vx355_all_gather.argtypes = [vp, vp, vp, sz]
    L.vx355_all_gather_v.argtypes = [vp, vp, P(i64), vp]
    L.vx355_exchange_create.argtypes = [vp, P(i32), i32, P(i32), i32, P(vp)]
    L.vx355_exchange_send.argtypes = [vp, P(abi.Batch)]
    L.vx355_exchange_receive.argtypes = [vp, P(abi.Column), P(i64)]
    L.vx355_exchange_destinations.argtypes = [vp, P(abi.Batch), i32, vp, i32]
    L.vx355_exchange_stream.restype = vp
    L.vx355_exchange_stream.argtypes = [vp]
    L.vx355_exchange_destroy.argtypes = [vp]
    L.vx355_exchange_destroy.restype = None
    L.vx355_join_repartition.argtypes = [vp, P(abi.JoinBuildSpec), P(abi.Batch), P(abi.JoinProbeSpec), P(abi.Batch),
                                         i32, JOIN_CHUNK_SINK, vp, P(vp)]
    L.vx355_agg_merge_partials.argtypes = [vp, vp, P(abi.AggSpec), P(vp)]
    L.vx355_value_dict_create.argtypes = [i32, i64, P(vp)]
    L.vx355_value_dict_compute.argtypes = [vp, P(abi.Batch), i32, vp, u64, vp, P(i32), i32]
    L.vx355_value_dict_lookup.argtypes = [vp, P(abi.Batch), i32, vp, u64, vp, vp, i32]
    L.vx355_value_dict_size.restype = i64
    L.vx355_value_dict_size.argtypes = [vp]
    L.vx355_value_dict_destroy.argtypes = [vp]
    L.vx355_value_dict_destroy.restype = None
    L.vx355_set_device.argtypes = [C.c_int]
    L.vx355_stream_wait_event.argtypes = [vp, vp]
    L.vx355_default_stream.restype = vp
    for name in ("vx355_agg_stream", "vx355_join_build_stream", "vx355_join_probe_stream"):
        getattr(L, name).restype = vp
        getattr(L, name).argtypes = [vp]
    _LIB = L
    return L


def _check(status):
    if status != abi.OK:
        raise Vx355Error(status, lib().vx355_last_error().decode())


def init(device=0):
    _check(lib().vx355_init(device))


def set_device(device):
    """vx355_set_device: the GPU of the calling thread (one Driver thread per device / rank)."""
    _check(lib().vx355_set_device(device))


def set_memory_limit(nbytes):
    """vx355_set_memory_limit: cap on the HBM the operators of this GPU hold at one time (0 = none)."""
    _check(lib().vx355_set_memory_limit(int(nbytes)))


def memory_usage():
    """vx355_memory_usage -> (bytes held by operators now, peak since the last call, bytes in the block cache)."""
    a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
    _check(lib().vx355_memory_usage(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def synchronize():
    _check(lib().vx355_synchronize())


class ValueDict:
    """VectorHasher in distinct-value mode (vx355_value_dict_*)."""

    def __init__(self, kind, range_size):
        h = C.c_void_p()
        _check(lib().vx355_value_dict_create(kind, range_size, C.byref(h)))
        self.h = h

    def compute(self, batch, col=0, rows=None, multiplier=1, result=None):
        n = batch.num_rows
        if result is None:
            result = np.zeros(max(1, n), dtype=np.uint64)
        bits = abi.pack_bits(rows) if rows is not None else None
        ok = C.c_int32()
        _check(lib().vx355_value_dict_compute(self.h, batch.ref(), col, bits.ctypes.data if bits is not None else None,
                                              multiplier, result.ctypes.data, C.byref(ok), abi.MEM_HOST))
        return bool(ok.value), result[:n]

    def lookup(self, batch, col=0, rows=None, multiplier=1, result=None):
        n = batch.num_rows
        if result is None:
            result = np.zeros(max(1, n), dtype=np.uint64)
        bits = abi.pack_bits(rows) if rows is not None else None
        out = np.zeros(max(1, (n + 63) // 64), dtype=np.uint64)
        _check(lib().vx355_value_dict_lookup(self.h, batch.ref(), col, bits.ctypes.data if bits is not None else None,
                                             multiplier, result.ctypes.data, out.ctypes.data, abi.MEM_HOST))
        return abi.unpack_bits(out, n), result[:n]

    def size(self):
        return lib().vx355_value_dict_size(self.h)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().vx355_value_dict_destroy(self.h)
                self.h = None
        except Exception:
            pass


# ---- multi-GPU exchange (RCCL inside the library) -----------------------------

class Comm:
    """vx355_comm: one rank's communicator, bound to the GPU of the calling thread."""

    def __init__(self, unique_id, world, rank):
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _check(lib().vx355_comm_create(buf, world, rank, C.byref(h)))
        self.h, self.world, self.rank = h, world, rank

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(lib().vx355_comm_get_unique_id(buf))
        return buf.raw

    @staticmethod
    def create_all(devices):
        """vx355_comm_create_all: ONE process, one communicator per entry of 'devices' (rank i on
        devices[i]; every device initialised with vx355_init) -> [Comm]. Each rank's collectives are then
        called from that rank's own thread (one process drives the node)."""
        n = len(devices)
        handles = (C.c_void_p * n)()
        _check(lib().vx355_comm_create_all(n, abi.i32_array(devices), handles))
        comms = []
        for rank in range(n):
            c = Comm.__new__(Comm)
            c.h, c.world, c.rank = C.c_void_p(handles[rank]), n, rank
            comms.append(c)
        return comms

    def exchange_counts(self, send_counts):
        send = (C.c_int64 * self.world)(*[int(x) for x in send_counts])
        recv = (C.c_int64 * self.world)()
        _check(lib().vx355_exchange_counts(self.h, send, recv))
        return list(recv)

    def exchange_columns(self, send_ptrs, widths, send_counts, recv_counts, recv_ptrs):
        n = len(send_ptrs)
        _check(lib().vx355_exchange_columns(self.h, (C.c_void_p * n)(*send_ptrs), abi.i32_array(widths), n,
                                            (C.c_int64 * self.world)(*[int(x) for x in send_counts]),
                                            (C.c_int64 * self.world)(*[int(x) for x in recv_counts]),
                                            (C.c_void_p * n)(*recv_ptrs)))

    def all_gather(self, send_ptr, recv_ptr, bytes_per_rank):
        _check(lib().vx355_all_gather(self.h, send_ptr, recv_ptr, bytes_per_rank))

    def all_gather_v(self, send_ptr, sizes, recv_ptr):
        _check(lib().vx355_all_gather_v(self.h, send_ptr, (C.c_int64 * self.world)(*[int(x) for x in sizes]), recv_ptr))

    def info(self):
        """(world, rank, device) as the RCCL communicator itself reports them."""
        w, r, d = C.c_int32(), C.c_int32(), C.c_int32()
        _check(lib().vx355_comm_info(self.h, C.byref(w), C.byref(r), C.byref(d)))
        return w.value, r.value, d.value

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().vx355_comm_destroy(self.h)
                self.h = None
        except Exception:
            pass


class Exchange:
    """vx355_exchange: one PartitionedOutput(keys) -> Exchange edge of a repartitioned join."""

    def __init__(self, comm, col_types, key_cols):
        h = C.c_void_p()
        _check(lib().vx355_exchange_create(comm.h, abi.i32_array(col_types), len(col_types), abi.i32_array(key_cols),
                                           len(key_cols), C.byref(h)))
        self.h, self.comm, self.col_types = h, comm, list(col_types)

    def send(self, batch):
        _check(lib().vx355_exchange_send(self.h, batch.ref()))

    def destinations(self, batch, num_destinations=0):
        """Destination rank per row for num_destinations ranks (0 = the communicator's size)."""
        out = np.zeros(max(1, batch.num_rows), dtype=np.uint32)
        _check(lib().vx355_exchange_destinations(self.h, batch.ref(), num_destinations, out.ctypes.data, abi.MEM_HOST))
        return out[: batch.num_rows]

    def receive(self):
        """-> (vx355_column array of FLAT device columns, rows); valid until the next receive."""
        cols = (abi.Column * len(self.col_types))()
        rows = C.c_int64()
        _check(lib().vx355_exchange_receive(self.h, cols, C.byref(rows)))
        return cols, rows.value

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().vx355_exchange_destroy(self.h)
                self.h = None
        except Exception:
            pass


def join_repartition(comm, build, build_batch, probe_spec_owner, probe_batch, chunks, sink):
    """vx355_join_repartition. build: (key_cols, key_types, dep_cols, dep_types, join_type) of the
    build side over build_batch's columns; probe_spec_owner: (key_cols, join_type) over
    probe_batch's; sink(chunk, received vx355_batch pointer, HashProbe) drains the probe.
    Returns the JoinTable."""
    key_cols, key_types, dep_cols, dep_types, join_type = build
    keep = [abi.i32_array(key_cols), abi.i32_array(key_types), abi.i32_array(dep_cols), abi.i32_array(dep_types)]
    bspec = abi.JoinBuildSpec(len(key_cols), keep[0], keep[1], len(dep_cols), keep[2], keep[3], join_type, 0, 0, 0)
    pkeys, pjoin = probe_spec_owner
    pk = abi.i32_array(pkeys)
    pspec = abi.JoinProbeSpec(len(pkeys), pk, pjoin, 0, 0, 0)
    table = JoinTable(None, list(dep_types))
    errors = []

    def trampoline(_arg, chunk, received, probe_handle):
        try:
            probe = HashProbe.__new__(HashProbe)   # borrowed handle: the library owns it
            probe.h, probe.table = None, table
            probe.borrowed = C.c_void_p(probe_handle)
            sink(chunk, received, probe)
            return abi.OK
        except Exception as e:  # noqa: BLE001 - must not unwind through the C frames
            errors.append(e)
            return abi.EINTERNAL
    cb = JOIN_CHUNK_SINK(trampoline)
    t = C.c_void_p()
    status = lib().vx355_join_repartition(comm.h, C.byref(bspec), build_batch.ref(), C.byref(pspec), probe_batch.ref(),
                                          chunks, cb, None, C.byref(t))
    if errors:
        raise errors[0]
    _check(status)
    table.t = t
    return table


def merge_partials(comm, partial, final_key_cols, final_key_types, final_aggs, step=abi.STEP_FINAL):
    """vx355_agg_merge_partials: -> HashAggregation (FINAL) that has consumed every rank's partial rows."""
    spec, keep = make_agg_spec(final_key_cols, final_key_types, final_aggs, step)
    h = C.c_void_p()
    _check(lib().vx355_agg_merge_partials(comm.h, partial.h, C.byref(spec), C.byref(h)))
    op = HashAggregation.__new__(HashAggregation)
    op.spec, op._keep, op.h = spec, keep, h
    types = (C.c_int32 * 64)()
    n = C.c_int32()
    _check(lib().vx355_agg_output_types(op.h, types, 64, C.byref(n)))
    op.kinds = list(types[: n.value])
    return op


# ---- device memory -------------------------------------------------------

class DeviceArray:
    """A typed HBM buffer owned through vx355_device_malloc."""

    def __init__(self, shape_or_array, dtype=None):
        if isinstance(shape_or_array, np.ndarray):
            a = np.ascontiguousarray(shape_or_array)
            self.dtype, self.shape = a.dtype, a.shape
            self.nbytes = a.nbytes
            self.ptr = self._alloc(self.nbytes)
            if self.nbytes:
                _check(lib().vx355_memcpy_h2d(self.ptr, a.ctypes.data, self.nbytes))
        else:
            self.dtype = np.dtype(dtype)
            self.shape = (shape_or_array,) if np.isscalar(shape_or_array) else tuple(shape_or_array)
            self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
            self.ptr = self._alloc(self.nbytes)

    @staticmethod
    def _alloc(nbytes):
        p = lib().vx355_device_malloc(max(64, nbytes + 64))
        if not p:
            raise Vx355Error(abi.ENOMEM, lib().vx355_last_error().decode())
        return p

    def zero(self):
        _check(lib().vx355_memset_d(self.ptr, 0, self.nbytes))
        return self

    def to_host(self, count=None):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            _check(lib().vx355_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes))
        return out if count is None else out.reshape(-1)[:count]

    def free(self):
        if getattr(self, "ptr", None):
            lib().vx355_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceColumn:
    """A decoded vector resident in HBM (vx355_column with mem = DEVICE)."""

    def __init__(self, host_col):
        self.kind, self.encoding = host_col.kind, host_col.encoding
        self.num_rows, self.base_size = host_col.num_rows, host_col.base_size
        values = host_col.values
        self.blob = None
        if host_col.kind in (abi.VARCHAR, abi.VARBINARY):
            raw = np.ascontiguousarray(host_col.values).view(np.uint8).reshape(-1, 16).copy()
            sizes = raw[:, 0:4].copy().view(np.uint32).reshape(-1)
            long_rows = np.nonzero(sizes > 12)[0]
            if len(long_rows):
                # bytes of the non-inline strings move to one HBM blob, the views point into it
                ptrs = raw[:, 8:16].copy().view(np.uint64).reshape(-1)
                parts = [C.string_at(int(ptrs[r]), int(sizes[r])) for r in long_rows]
                self.blob = DeviceArray(np.frombuffer(b"".join(parts), dtype=np.uint8).copy())
                at = 0
                for r, part in zip(long_rows, parts):
                    ptrs[r] = self.blob.ptr + at
                    at += len(part)
                raw[:, 8:16] = ptrs.view(np.uint8).reshape(-1, 8)
            values = raw
        self.values = DeviceArray(values)
        self.nulls = DeviceArray(host_col.nulls) if host_col.nulls is not None else None
        self.indices = DeviceArray(host_col.indices) if host_col.indices is not None else None

    @classmethod
    def from_ptr(cls, kind, ptr, num_rows, nulls_ptr=None, encoding=abi.FLAT, indices_ptr=None,
                 base_size=0):
        """Aliases memory owned elsewhere (e.g. a torch tensor's data_ptr())."""
        self = cls.__new__(cls)
        self.kind, self.encoding, self.num_rows, self.base_size = kind, encoding, num_rows, base_size
        self.values = self.nulls = self.indices = None
        self._ptrs = (ptr, nulls_ptr, indices_ptr)
        return self

    def descriptor(self):
        c = abi.Column()
        c.type_kind, c.encoding = self.kind, self.encoding
        if self.values is None:
            c.values, c.nulls, c.indices = self._ptrs
        else:
            c.values = self.values.ptr
            c.nulls = self.nulls.ptr if self.nulls is not None else None
            c.indices = self.indices.ptr if self.indices is not None else None
        c.base_size = self.base_size if self.encoding == abi.DICTIONARY else 0
        c.mem = abi.MEM_DEVICE
        return c


def to_device(host_batch):
    """HostBatch -> HostBatch-like object whose columns live in HBM."""
    return abi.HostBatch([DeviceColumn(c) for c in host_batch.columns], host_batch.num_rows)


def compose_indices_device(inner_ptr, inner_size, outer_ptr, n, out_ptr):
    """out[i] = inner[outer[i]] on HBM-resident int32 arrays (dictionary over a dictionary)."""
    _check(lib().vx355_compose_indices(C.c_void_p(inner_ptr), inner_size, C.c_void_p(outer_ptr), n, C.c_void_p(out_ptr),
                                       abi.MEM_DEVICE))


def compose_indices(inner, outer):
    """Host arrays: numpy int32 inner[outer]."""
    inner = np.ascontiguousarray(inner, dtype=np.int32)
    outer = np.ascontiguousarray(outer, dtype=np.int32)
    out = np.empty(len(outer), dtype=np.int32)
    _check(lib().vx355_compose_indices(inner.ctypes.data_as(C.c_void_p), len(inner), outer.ctypes.data_as(C.c_void_p),
                                       len(outer), out.ctypes.data_as(C.c_void_p), abi.MEM_HOST))
    return out


def hbm_ceiling(kind, nbytes=4 << 30, iterations=5):
    """GB/s of the library's read-only stream (abi.CEILING_READ) or copy (abi.CEILING_COPY) kernel."""
    out = C.c_double()
    _check(lib().vx355_hbm_ceiling(kind, C.c_size_t(nbytes), iterations, C.byref(out)))
    return out.value


# ---- profiling -----------------------------------------------------------

def profile_enable(on=True):
    _check(lib().vx355_profile_enable(1 if on else 0))


def profile_reset():
    _check(lib().vx355_profile_reset())


def profile():
    """-> {kernel name: (total_ms, launches)} since the last reset."""
    buf = C.create_string_buffer(1 << 16)
    _check(lib().vx355_profile_names(buf, len(buf)))
    out = {}
    for name in filter(None, buf.value.decode().split("\n")):
        ms, n = C.c_double(), C.c_int64()
        _check(lib().vx355_profile_get(name.encode(), C.byref(ms), C.byref(n)))
        out[name] = (ms.value, n.value)
    return out


# ---- standalone kernels (the parity tests call these and the checker alike) --

def hash_columns(batch, key_cols, rows=None, mix_first=False, out=None):
    n = batch.num_rows
    if out is None:
        out = np.zeros(max(1, n), dtype=np.uint64)
    bits = abi.pack_bits(rows) if rows is not None else None
    _check(lib().vx355_hash_columns(batch.ref(), abi.i32_array(key_cols), len(key_cols),
                                    bits.ctypes.data if bits is not None else None,
                                    1 if mix_first else 0, out.ctypes.data, abi.MEM_HOST))
    return out[:n]


def value_ids(batch, key_cols, specs, rows=None, lookup=False, result=None):
    n = batch.num_rows
    if result is None:
        result = np.zeros(max(1, n), dtype=np.uint64)
    bits = abi.pack_bits(rows) if rows is not None else None
    rows_out = np.zeros(max(1, abi.num_words(n)), dtype=np.uint64)
    mapped = C.c_int32(1)
    arr = (abi.ValueIdSpec * max(1, len(specs)))(*[abi.ValueIdSpec(*s) for s in specs])
    _check(lib().vx355_value_ids(batch.ref(), abi.i32_array(key_cols), arr, len(key_cols),
                                 bits.ctypes.data if bits is not None else None,
                                 1 if lookup else 0, result.ctypes.data, rows_out.ctypes.data,
                                 C.byref(mapped), abi.MEM_HOST))
    return result[:n], abi.unpack_bits(rows_out, n), bool(mapped.value)


def filter_compact(values, nulls=None, rows=None):
    n = len(values)
    v = abi.pack_bits(values)
    nl = abi.pack_bits(nulls) if nulls is not None else None
    rw = abi.pack_bits(rows) if rows is not None else None
    out = np.zeros(max(1, n), dtype=np.int32)
    cnt = C.c_int32()
    _check(lib().vx355_filter_compact(v.ctypes.data, nl.ctypes.data if nl is not None else None,
                                      rw.ctypes.data if rw is not None else None, n,
                                      out.ctypes.data, C.byref(cnt), abi.MEM_HOST))
    return out[: cnt.value].copy()


def filter_compact_device(values_ptr, num_rows, idx_out_ptr, nulls_ptr=None, rows_ptr=None):
    """All buffers in HBM; returns the number of selected rows."""
    cnt = C.c_int32()
    _check(lib().vx355_filter_compact(values_ptr, nulls_ptr, rows_ptr, num_rows, idx_out_ptr,
                                      C.byref(cnt), abi.MEM_DEVICE))
    return cnt.value


def partition_scatter(partitions, num_partitions, cols):
    """Stable partition of numpy columns (host buffers) -> (reordered columns, counts)."""
    partitions = np.ascontiguousarray(partitions, dtype=np.uint32)
    n = len(partitions)
    cols = [np.ascontiguousarray(c) for c in cols]
    outs = [np.empty_like(c) for c in cols]
    widths = abi.i32_array([c.dtype.itemsize * (c.shape[1] if c.ndim == 2 else 1) for c in cols])
    ins = (C.c_void_p * max(1, len(cols)))(*[c.ctypes.data for c in cols])
    outp = (C.c_void_p * max(1, len(cols)))(*[o.ctypes.data for o in outs])
    counts = (C.c_int64 * num_partitions)()
    _check(lib().vx355_partition_scatter(partitions.ctypes.data, n, num_partitions, ins, widths,
                                         len(cols), outp, counts, abi.MEM_HOST))
    return outs, np.array(list(counts), dtype=np.int64)


def partition_scatter_device(partitions_ptr, num_rows, num_partitions, in_ptrs, widths, out_ptrs):
    """Device-resident variant; returns the per-partition row counts."""
    ins = (C.c_void_p * max(1, len(in_ptrs)))(*in_ptrs)
    outp = (C.c_void_p * max(1, len(out_ptrs)))(*out_ptrs)
    counts = (C.c_int64 * num_partitions)()
    _check(lib().vx355_partition_scatter(partitions_ptr, num_rows, num_partitions, ins,
                                         abi.i32_array(widths), len(in_ptrs), outp, counts,
                                         abi.MEM_DEVICE))
    return np.array(list(counts), dtype=np.int64)


def filter_project(batch, terms, projs, with_nulls=False):
    """FilterProject on host buffers -> (selected row numbers, [projection values],
    [projection validity or None])."""
    n = batch.num_rows
    idx = np.zeros(max(1, n), dtype=np.int32)
    outs = [np.zeros(max(1, n), dtype=np.float64) for _ in projs]
    nulls = [np.zeros(max(1, abi.num_words(n)), dtype=np.uint64) for _ in projs] if with_nulls else None
    cnt = C.c_int32()
    out_ptrs = (C.c_void_p * max(1, len(projs)))(*[o.ctypes.data for o in outs])
    null_ptrs = (C.c_void_p * max(1, len(projs)))(*[x.ctypes.data for x in nulls]) if with_nulls else None
    _check(lib().vx355_filter_project(batch.ref(), abi.filter_terms(terms), len(terms),
                                      abi.projections(projs), len(projs), idx.ctypes.data,
                                      C.byref(cnt), out_ptrs, null_ptrs, abi.MEM_HOST))
    m = cnt.value
    return (idx[:m].copy(), [o[:m].copy() for o in outs],
            [abi.unpack_bits(x, m) for x in nulls] if with_nulls else None)


def filter_project_device(batch, terms, projs, idx_ptr, proj_ptrs, null_ptrs=None):
    """Everything resident in HBM; returns the number of selected rows."""
    cnt = C.c_int32()
    out_ptrs = (C.c_void_p * max(1, len(projs)))(*proj_ptrs)
    nptrs = (C.c_void_p * max(1, len(projs)))(*null_ptrs) if null_ptrs else None
    _check(lib().vx355_filter_project(batch.ref(), abi.filter_terms(terms), len(terms),
                                      abi.projections(projs), len(projs), idx_ptr, C.byref(cnt),
                                      out_ptrs, nptrs, abi.MEM_DEVICE))
    return cnt.value


def partition(hashes, kind, num_partitions=0, bit_begin=0, bit_end=0):
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    out = np.zeros(max(1, len(hashes)), dtype=np.uint32)
    _check(lib().vx355_partition(hashes.ctypes.data, len(hashes), kind, num_partitions, bit_begin,
                                 bit_end, out.ctypes.data, abi.MEM_HOST))
    return out[: len(hashes)]


def presto_serialize(batch, offsets, rows=None, flags=0, device_out=False):
    """PartitionedOutput's pages (vx355_presto_serialize): -> list of bytes objects, one per
    row range [offsets[p], offsets[p + 1]) of rows (or of the batch rows when rows is None).
    device_out: the pages are written to an HBM buffer (and fetched from there)."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    num_pages = len(offsets) - 1
    rows_ptr = None
    if rows is not None:
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        rows_ptr = rows.ctypes.data
    page_offsets = np.zeros(num_pages + 1, dtype=np.int64)
    _check(lib().vx355_presto_serialize(batch.ref(), rows_ptr, abi.MEM_HOST, offsets.ctypes.data, num_pages, flags,
                                        None, 0, abi.MEM_HOST, page_offsets.ctypes.data))
    total = int(page_offsets[-1])
    out = np.zeros(max(total, 1), dtype=np.uint8)
    again = np.zeros(num_pages + 1, dtype=np.int64)
    if device_out:
        dev = DeviceArray(max(total, 1), np.uint8)
        _check(lib().vx355_presto_serialize(batch.ref(), rows_ptr, abi.MEM_HOST, offsets.ctypes.data, num_pages,
                                            flags, dev.ptr, total, abi.MEM_DEVICE, again.ctypes.data))
        if total:
            _check(lib().vx355_memcpy_d2h(out.ctypes.data, dev.ptr, total))
    else:
        _check(lib().vx355_presto_serialize(batch.ref(), rows_ptr, abi.MEM_HOST, offsets.ctypes.data, num_pages,
                                            flags, out.ctypes.data, total, abi.MEM_HOST, again.ctypes.data))
    assert (again == page_offsets).all()
    return [out[page_offsets[p]:page_offsets[p + 1]].tobytes() for p in range(num_pages)]


def presto_compress_page(page, compression, min_ratio=0.8):
    """vx355_presto_compress_page (host work, no GPU): one uncompressed page -> the page a compressing
    exchange writes (the same bytes when compression does not reach min_ratio)."""
    page = bytes(page)
    out = C.create_string_buffer(max(len(page), 1))
    size = C.c_int64()
    _check(lib().vx355_presto_compress_page(page, len(page), compression, min_ratio, out, len(page), C.byref(size)))
    return out.raw[:size.value]


def presto_uncompress_page(page, compression):
    """vx355_presto_uncompress_page (host work, no GPU): -> the uncompressed page."""
    import struct
    page = bytes(page)
    cap = 21 + max(struct.unpack_from("<i", page, 5)[0], 0) if len(page) >= 21 else 21
    out = C.create_string_buffer(max(cap, len(page), 1))
    size = C.c_int64()
    _check(lib().vx355_presto_uncompress_page(page, len(page), compression, out, len(out), C.byref(size)))
    return out.raw[:size.value]


def presto_deserialize(pages, kinds, flags=0):
    """vx355_presto_deserialize: list of page bytes -> [(values, valid)] per column, fetched back
    from the HBM columns the library wrote (strings longer than 12 bytes through the device copy
    of the pages their views point into). A struct column is described as ("row", [field kinds])
    and comes back as ([(values, valid) per field], struct valid)."""
    import struct
    pages = [bytes(p) for p in pages if len(p)]
    total_rows = sum(struct.unpack_from("<i", p, 0)[0] for p in pages)
    keep = [C.create_string_buffer(p, len(p)) for p in pages]
    ptrs = (C.c_void_p * max(1, len(pages)))(*[C.addressof(k) for k in keep])
    sizes = np.array([len(p) for p in pages] or [0], dtype=np.int64)
    # (a compressed page - codec marker bit 1 - occupies 21 + uncompressedSize bytes of the device buffer)
    total_bytes = int(sum(21 + struct.unpack_from("<i", p, 5)[0] if len(p) >= 21 and (p[4] & 1) else len(p) for p in pages))
    dev_bytes = DeviceArray(max(total_bytes, 1), np.uint8)
    cap = max(total_rows, 1)
    words = (cap + 63) // 64
    # nodes in prefix order: a struct's own entry (validity only), then its fields
    nodes, types = [], []
    for kind in kinds:
        if isinstance(kind, tuple):
            nodes.append(abi.ROW)
            types.append(abi.ROW | (len(kind[1]) << 8))   # VX355_ROW_OF(number of fields)
            nodes.extend(kind[1])
            types.extend(kind[1])
        else:
            nodes.append(kind)
            types.append(kind)
    vals, nulls = [], []
    descs = (abi.OutColumn * max(1, len(nodes)))()
    for c, kind in enumerate(nodes):
        width = {abi.BOOLEAN: 0, abi.TINYINT: 1, abi.SMALLINT: 2, abi.INTEGER: 4, abi.REAL: 4, abi.BIGINT: 8,
                 abi.DOUBLE: 8, abi.TIMESTAMP: 16, abi.VARCHAR: 16, abi.VARBINARY: 16, abi.ROW: 0}[kind]
        vals.append(DeviceArray(words * 8 if width == 0 else cap * width, np.uint8))
        nulls.append(DeviceArray(words, np.uint64))
        descs[c].type_kind, descs[c].mem = kind, abi.MEM_DEVICE
        descs[c].values, descs[c].nulls = (None if kind == abi.ROW else vals[c].ptr), nulls[c].ptr
    rows = C.c_int64()
    _check(lib().vx355_presto_deserialize(ptrs, sizes.ctypes.data, len(pages), abi.i32_array(types), len(nodes), flags,
                                          dev_bytes.ptr, total_bytes, descs, cap, C.byref(rows)))
    n = rows.value
    assert n == total_rows
    host_bytes = dev_bytes.to_host().tobytes() if total_bytes else b""

    def fetch(c, kind):
        valid = abi.unpack_bits(nulls[c].to_host(), n)
        if kind == abi.ROW:
            return None, valid
        raw = vals[c].to_host()
        if kind == abi.BOOLEAN:
            v = abi.unpack_bits(raw.view(np.uint64), n)
        elif kind in (abi.VARCHAR, abi.VARBINARY):
            views = raw[: n * 16].reshape(n, 16)
            v = []
            for r in range(n):
                size = int(views[r, 0:4].view(np.uint32)[0])
                if size <= 12:
                    v.append(views[r, 4:4 + size].tobytes())
                else:
                    at = int(views[r, 8:16].view(np.uint64)[0]) - dev_bytes.ptr
                    assert 0 <= at and at + size <= total_bytes
                    assert host_bytes[at:at + 4] == views[r, 4:8].tobytes()  # the prefix
                    v.append(host_bytes[at:at + size])
        elif kind == abi.TIMESTAMP:
            v = raw[: n * 16].view(np.int64).reshape(n, 2)
        else:
            v = raw[: n * np.dtype(abi.KIND_DTYPE[kind]).itemsize].view(abi.KIND_DTYPE[kind])
        return v, valid
    out, c = [], 0
    for kind in kinds:
        if isinstance(kind, tuple):
            _, struct_valid = fetch(c, abi.ROW)
            fields = [fetch(c + 1 + f, k) for f, k in enumerate(kind[1])]
            out.append((fields, struct_valid))
            c += 1 + len(kind[1])
        else:
            out.append(fetch(c, kind))
            c += 1
    return n, out


# ---- HashAggregation -------------------------------------------------------

def make_agg_spec(key_cols, key_types, aggs, step, ignore_null_keys=False, flags=0):
    """aggs: list of (kind, input_col, input_type[, mask_col[, input_col2]])."""
    keep = {"kc": abi.i32_array(key_cols), "kt": abi.i32_array(key_types)}
    fns = (abi.AggFn * max(1, len(aggs)))()
    for i, a in enumerate(aggs):
        kind, col, typ = a[0], a[1], a[2]
        mask = a[3] if len(a) > 3 else -1
        col2 = a[4] if len(a) > 4 else -1
        fn_flags = a[5] if len(a) > 5 else 0
        fns[i] = abi.AggFn(kind, col, col2, typ, mask, fn_flags)
    keep["fns"] = fns
    spec = abi.AggSpec(len(key_cols), keep["kc"], keep["kt"], len(aggs), fns, step,
                       1 if ignore_null_keys else 0, flags, 0)
    keep["spec"] = spec
    return spec, keep


class HashAggregation:
    """exec::HashAggregation (exec/HashAggregation.h) on the MI355X."""

    def __init__(self, key_cols, key_types, aggs, step=abi.STEP_SINGLE, ignore_null_keys=False, flags=0):
        self.spec, self._keep = make_agg_spec(key_cols, key_types, aggs, step, ignore_null_keys, flags)
        h = C.c_void_p()
        _check(lib().vx355_agg_create(C.byref(self.spec), C.byref(h)))
        self.h = h
        types = (C.c_int32 * 64)()
        n = C.c_int32()
        _check(lib().vx355_agg_output_types(self.h, types, 64, C.byref(n)))
        self.kinds = list(types[: n.value])

    def __del__(self):
        try:  # at interpreter shutdown module globals may already be gone
            if getattr(self, "h", None):
                lib().vx355_agg_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_fused_input(self, terms, projs):
        """Fuse the upstream FilterProject: add_input then takes its INPUT batches;
        aggregate input PROJ(j) reads projection j."""
        _check(lib().vx355_agg_set_fused_input(self.h, abi.filter_terms(terms), len(terms),
                                               abi.projections(projs), len(projs)))

    def add_input(self, batch):
        _check(lib().vx355_agg_add_input(self.h, batch.ref()))

    # asynchronous boundary: the batch (and its buffers) is kept alive here until wait() / poll()
    # reports its ticket completed
    def add_input_async(self, batch):
        ticket = C.c_int64()
        _check(lib().vx355_agg_add_input_async(self.h, batch.ref(), C.byref(ticket)))
        self.__dict__.setdefault("_in_flight", {})[ticket.value] = batch
        return ticket.value

    def poll(self):
        sub, done = C.c_int64(), C.c_int64()
        _check(lib().vx355_agg_poll(self.h, C.byref(sub), C.byref(done)))
        held = self.__dict__.get("_in_flight", {})
        for t in [t for t in held if t <= done.value]:
            del held[t]
        return sub.value, done.value

    def wait(self):
        status = lib().vx355_agg_wait(self.h)
        self.__dict__.get("_in_flight", {}).clear()
        _check(status)

    def no_more_input(self):
        _check(lib().vx355_agg_no_more_input(self.h))

    # queued forms: neither call waits for the batches queued before it
    def no_more_input_async(self):
        ticket = C.c_int64()
        _check(lib().vx355_agg_no_more_input_async(self.h, C.byref(ticket)))
        return ticket.value

    def get_output_async(self, max_rows=1024, done=None):
        """One page queued behind everything submitted so far; 'done(status, num_rows, finished)' runs on the
        library's worker thread. Returns (ticket, buffers): output_result(ticket, buffers) once poll() says so."""
        out = abi.OutBuffers(self.kinds, max_rows)
        ticket = C.c_int64()
        cb = OUTPUT_DONE_FN(lambda _arg, status, n, fin: done(status, n, bool(fin))) if done else None
        self.__dict__.setdefault("_page_callbacks", []).append(cb)  # (alive until the handle goes)
        _check(lib().vx355_agg_get_output_async(self.h, out.descs, len(self.kinds), max_rows, cb, None,
                                                C.byref(ticket)))
        return ticket.value, out

    def output_result(self, ticket, out):
        n, fin = C.c_int32(), C.c_int32()
        _check(lib().vx355_agg_output_result(self.h, C.c_int64(ticket), C.byref(n), C.byref(fin)))
        return [out.column(i, n.value) for i in range(len(self.kinds))], n.value, bool(fin.value)

    def flush(self):
        """Partial flush: get_output then drains the groups so far; the table restarts empty."""
        _check(lib().vx355_agg_flush(self.h))

    def to_intermediate(self, batch, kinds):
        """GroupingSet::toIntermediate: the aggregate columns of the PARTIAL layout for the raw
        rows of 'batch' (kinds = their types, e.g. dist.partial_kinds(...)[num_keys:])."""
        n = batch.num_rows
        out = abi.OutBuffers(kinds, n)
        _check(lib().vx355_agg_to_intermediate(self.h, batch.ref(), out.descs, len(kinds)))
        return [out.column(i, n) for i in range(len(kinds))]

    def get_output(self, max_rows=1024):
        out = abi.OutBuffers(self.kinds, max_rows)
        n, fin = C.c_int32(), C.c_int32()
        _check(lib().vx355_agg_get_output(self.h, out.descs, len(self.kinds), max_rows,
                                          C.byref(n), C.byref(fin)))
        return [out.column(i, n.value) for i in range(len(self.kinds))], n.value, bool(fin.value)

    def stats(self):
        s = abi.AggStats()
        _check(lib().vx355_agg_get_stats(self.h, C.byref(s)))
        return s


Aggregation = HashAggregation
PROJECTION_COL_BASE = 1 << 20


def PROJ(j):
    return PROJECTION_COL_BASE + j


def collect_output(op, max_rows=1024):
    """Drain get_output: -> list of columns, each (values, valid) concatenated."""
    cols = None
    while True:
        batch, n, fin = op.get_output(max_rows)
        if cols is None:
            cols = [([], []) for _ in batch]
        for i, (vals, valid) in enumerate(batch):
            cols[i][0].append(vals)
            cols[i][1].append(valid)
        if fin:
            break
    out = []
    for vals, valid in cols:
        if vals and isinstance(vals[0], list):
            v = [x for part in vals for x in part]
        else:
            v = np.concatenate(vals) if vals else np.zeros(0)
        out.append((v, np.concatenate(valid) if valid else np.zeros(0, bool)))
    return out


# ---- HashBuild / HashProbe ---------------------------------------------------

class HashBuild:
    """exec::HashBuild (exec/HashBuild.h): one per build Driver."""

    def __init__(self, key_cols, key_types, dep_cols=(), dep_types=(), join_type=abi.JOIN_INNER, null_aware=False,
                 null_as_value=False, drop_duplicates=False):
        self._keep = [abi.i32_array(key_cols), abi.i32_array(key_types), abi.i32_array(dep_cols),
                      abi.i32_array(dep_types)]
        self.spec = abi.JoinBuildSpec(len(key_cols), self._keep[0], self._keep[1], len(dep_cols),
                                      self._keep[2], self._keep[3], join_type, 1 if null_aware else 0,
                                      1 if null_as_value else 0, 1 if drop_duplicates else 0)
        self.dep_types = list(dep_types)
        h = C.c_void_p()
        _check(lib().vx355_join_build_create(C.byref(self.spec), C.byref(h)))
        self.h = h

    def add_input(self, batch):
        _check(lib().vx355_join_build_add_input(self.h, batch.ref()))

    def add_input_async(self, batch):
        """Queues the batch for the handle's worker thread (kept alive here until wait())."""
        ticket = C.c_int64()
        _check(lib().vx355_join_build_add_input_async(self.h, batch.ref(), C.byref(ticket)))
        self.__dict__.setdefault("_in_flight", {})[ticket.value] = batch
        return ticket.value

    def poll(self):
        sub, done = C.c_int64(), C.c_int64()
        _check(lib().vx355_join_build_poll(self.h, C.byref(sub), C.byref(done)))
        held = self.__dict__.get("_in_flight", {})
        for t in [t for t in held if t <= done.value]:   # completed batches are no longer kept alive
            del held[t]
        return sub.value, done.value

    def wait(self):
        status = lib().vx355_join_build_wait(self.h)
        self.__dict__.get("_in_flight", {}).clear()
        _check(status)

    def finish(self, others=()):
        arr = (C.c_void_p * max(1, len(others)))(*[o.h for o in others])
        t = C.c_void_p()
        _check(lib().vx355_join_build_finish(self.h, arr, len(others), C.byref(t)))
        self.__dict__.get("_in_flight", {}).clear()   # finish waited for the queue
        return JoinTable(t, self.dep_types)

    def __del__(self):
        try:  # at interpreter shutdown module globals may already be gone
            if getattr(self, "h", None):
                lib().vx355_join_build_destroy(self.h)
                self.h = None
        except Exception:
            pass


JoinBuild = HashBuild


class JoinTable:
    """The finished table HashJoinBridge hands from build to probe."""

    def __init__(self, t, dep_types):
        self.t = t
        self.dep_types = dep_types

    def stats(self):
        s = abi.JoinTableStats()
        _check(lib().vx355_join_table_get_stats(self.t, C.byref(s)))
        return s

    # Dynamic filters for the probe-side scan (HashProbe::pushdownDynamicFilters).
    def key_filter(self, key):
        f = abi.KeyFilter()
        _check(lib().vx355_join_table_key_filter(self.t, key, C.byref(f)))
        return f

    def key_filter_values(self, key):
        f = self.key_filter(key)
        out = np.zeros(max(1, f.num_distinct), dtype=np.int64)
        n = C.c_int64()
        _check(lib().vx355_join_table_key_filter_values(self.t, key, out.ctypes.data, len(out), abi.MEM_HOST,
                                                        C.byref(n)))
        return out[:n.value]

    def key_filter_bloom(self, key, lanes=8, false_positive=0.01):
        """(blocks as uint32[num_blocks, lanes]) sized like BigintValuesUsingBloomFilter::numBlocks."""
        f = self.key_filter(key)
        nb = lib().vx355_bloom_num_blocks(max(1, f.num_distinct), false_positive, lanes)
        blocks = np.zeros((nb, lanes), dtype=np.uint32)
        _check(lib().vx355_join_table_key_filter_bloom(self.t, key, lanes, blocks.ctypes.data, nb, abi.MEM_HOST))
        return blocks

    def __del__(self):
        try:  # at interpreter shutdown module globals may already be gone
            if getattr(self, "t", None):
                lib().vx355_join_table_release(self.t)
                self.t = None
        except Exception:
            pass


def bloom_test(blocks, column, rows=None):
    """BigintValuesUsingBloomFilter::testInt64 over a HostColumn -> bool array."""
    n = column.num_rows
    words = (n + 63) // 64
    out = np.zeros(max(1, words), dtype=np.uint64)
    rows_ptr = None
    if rows is not None:
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        rows_ptr = rows.ctypes.data
    desc = column.descriptor()
    blocks = np.ascontiguousarray(blocks, dtype=np.uint32)
    _check(lib().vx355_bloom_test(blocks.ctypes.data, blocks.shape[0], blocks.shape[1], C.byref(desc), n,
                                  rows_ptr, out.ctypes.data, abi.MEM_HOST))
    return np.unpackbits(out.view(np.uint8), bitorder="little")[:n].astype(bool)


class HashProbe:
    """exec::HashProbe (exec/HashProbe.h)."""

    def __init__(self, table, key_cols, join_type=abi.JOIN_INNER, null_aware=False, null_as_value=False):
        self.table = table
        self._keep = abi.i32_array(key_cols)
        self.spec = abi.JoinProbeSpec(len(key_cols), self._keep, join_type, 1 if null_aware else 0,
                                      1 if null_as_value else 0, 0)
        h = C.c_void_p()
        _check(lib().vx355_join_probe_create(table.t, C.byref(self.spec), C.byref(h)))
        self.h = h

    def get_build_side_output(self, max_rows=1024, build_col_ids=None):
        """HashProbe::getBuildSideOutput (right / full / right semi): -> (build rows, columns, finished)."""
        if build_col_ids is None:
            build_col_ids = list(range(len(self.table.dep_types)))
        kinds = [abi.BOOLEAN if i == abi.BUILD_COL_MATCH else self.table.dep_types[i] for i in build_col_ids]
        out = abi.OutBuffers(kinds, max_rows)
        build_rows = np.zeros(max(1, max_rows), dtype=np.int32)
        n, fin = C.c_int32(), C.c_int32()
        ids = abi.i32_array(build_col_ids)
        _check(lib().vx355_join_probe_get_build_side_output(self.h, max_rows, build_rows.ctypes.data, abi.MEM_HOST,
                                                            out.descs, ids, len(kinds), C.byref(n), C.byref(fin)))
        cols = [out.column(i, n.value) for i in range(len(kinds))]
        return build_rows[: n.value].copy(), cols, bool(fin.value)

    def set_filter(self, terms):
        """HashJoinNode::filter: [(left, cmp, right)], see abi.join_filter_terms."""
        self._filter = abi.join_filter_terms(terms)
        _check(lib().vx355_join_probe_set_filter(self.h, self._filter, len(terms)))

    def set_output_batch_bytes(self, nbytes):
        _check(lib().vx355_join_probe_set_output_batch_bytes(self.h, nbytes))

    def set_input_filter(self, terms):
        """Fuse the filter-only FilterProject in front of the probe: add_input then takes ITS input
        batches, rows failing [(col, cmp, constant)] find nothing, mappings number the unfiltered rows."""
        _check(lib().vx355_join_probe_set_input_filter(self.h, abi.filter_terms(terms), len(terms)))

    def add_input(self, batch):
        self._batch = batch
        _check(lib().vx355_join_probe_add_input(self.h, batch.ref()))

    def add_input_regrouped(self, batch, out_ptrs):
        """vx355_join_probe_add_input_regrouped: out_ptrs = one device pointer per column of the batch
        (num_rows x width bytes each). -> True when the operator's input batch now is the regrouped
        copy in out_ptrs (mappings number ITS rows), False when the batch was probed as it came."""
        self._batch = batch
        self._regrouped_ptrs = (C.c_void_p * len(out_ptrs))(*out_ptrs)
        moved = C.c_int32()
        _check(lib().vx355_join_probe_add_input_regrouped(self.h, batch.ref(), self._regrouped_ptrs, C.byref(moved)))
        return bool(moved.value)

    def add_input_async(self, batch):
        """-> ticket; the batch is kept alive here. get_output (or wait) waits for it."""
        self._batch = batch
        ticket = C.c_int64()
        _check(lib().vx355_join_probe_add_input_async(self.h, batch.ref(), C.byref(ticket)))
        return ticket.value

    def poll(self):
        sub, done = C.c_int64(), C.c_int64()
        _check(lib().vx355_join_probe_poll(self.h, C.byref(sub), C.byref(done)))
        return sub.value, done.value

    def wait(self):
        _check(lib().vx355_join_probe_wait(self.h))

    def get_output(self, max_rows=1024, build_col_ids=None):
        if build_col_ids is None:
            build_col_ids = list(range(len(self.table.dep_types)))
        kinds = [self.table.dep_types[i] for i in build_col_ids]
        out = abi.OutBuffers(kinds, max_rows)
        mapping = np.zeros(max(1, max_rows), dtype=np.int32)
        build_rows = np.zeros(max(1, max_rows), dtype=np.int32)
        n, fin = C.c_int32(), C.c_int32()
        ids = abi.i32_array(build_col_ids)
        _check(lib().vx355_join_probe_get_output(self.h, max_rows, mapping.ctypes.data,
                                                 build_rows.ctypes.data, abi.MEM_HOST, out.descs,
                                                 ids, len(kinds), C.byref(n), C.byref(fin)))
        cols = [out.column(i, n.value) for i in range(len(kinds))]
        return mapping[: n.value].copy(), build_rows[: n.value].copy(), cols, bool(fin.value)

    def get_output_async(self, max_rows=1024, build_col_ids=None, done=None, build_side=False):
        """One page of output queued behind the batch of add_input_async; -> (ticket, page): output_result(ticket,
        page) once poll() reports it complete. 'done(status, rows, finished)' runs on the library's worker."""
        if build_col_ids is None:
            build_col_ids = list(range(len(self.table.dep_types)))
        kinds = [self.table.dep_types[i] for i in build_col_ids]
        page = {"out": abi.OutBuffers(kinds, max_rows), "mapping": np.zeros(max(1, max_rows), dtype=np.int32),
                "build_rows": np.zeros(max(1, max_rows), dtype=np.int32), "ids": abi.i32_array(build_col_ids),
                "kinds": kinds}
        page["cb"] = OUTPUT_DONE_FN(lambda _arg, status, n, fin: done(status, n, bool(fin))) if done else None
        ticket = C.c_int64()
        _check(lib().vx355_join_probe_get_output_async(self.h, 1 if build_side else 0, max_rows, page["mapping"].ctypes.data,
                                                       page["build_rows"].ctypes.data, abi.MEM_HOST, page["out"].descs,
                                                       page["ids"], len(kinds), page["cb"], None, C.byref(ticket)))
        self.__dict__.setdefault("_pages", {})[ticket.value] = page   # (buffers and callback stay alive)
        return ticket.value, page

    def output_result(self, ticket, page):
        n, fin = C.c_int32(), C.c_int32()
        status = lib().vx355_join_probe_output_result(self.h, C.c_int64(ticket), C.byref(n), C.byref(fin))
        _check(status)
        self.__dict__.get("_pages", {}).pop(ticket, None)
        cols = [page["out"].column(i, n.value) for i in range(len(page["kinds"]))]
        return page["mapping"][: n.value].copy(), page["build_rows"][: n.value].copy(), cols, bool(fin.value)

    def get_output_device(self, max_rows, mapping_ptr, build_rows_ptr, out_descs=None,
                          build_col_ids=()):
        """Device-resident outputs; -> (n, finished)."""
        n, fin = C.c_int32(), C.c_int32()
        ids = abi.i32_array(list(build_col_ids))
        _check(lib().vx355_join_probe_get_output(self.h or self.borrowed, max_rows, mapping_ptr, build_rows_ptr,
                                                 abi.MEM_DEVICE, out_descs, ids,
                                                 len(build_col_ids), C.byref(n), C.byref(fin)))
        return n.value, bool(fin.value)

    def __del__(self):
        try:  # at interpreter shutdown module globals may already be gone
            if getattr(self, "h", None):
                lib().vx355_join_probe_destroy(self.h)
                self.h = None
        except Exception:
            pass


JoinProbe = HashProbe
