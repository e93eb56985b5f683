"""TPC-H Q3 assembled from the library's operators (the reference's plan:
exec/tests/utils/TpchQueryBuilder.cpp:467-558), everything resident in HBM:

  customer -> FilterProject(c_mktsegment = 'BUILDING') -> HashBuild(c_custkey)
  orders   -> FilterProject(o_orderdate < DATE) -> HashProbe(o_custkey)
           -> HashBuild(o_orderkey; payload o_orderdate, o_shippriority)
  lineitem -> FilterProject(l_shipdate > DATE) -> HashProbe(l_orderkey)
           -> HashAggregation(l_orderkey, o_orderdate, o_shippriority;
                              sum(l_extendedprice * (1 - l_discount)))

Pass-through columns travel as dictionary vectors over the selected-row
indices, exactly like FilterProject / HashProbe::fillOutput wrap them
(exec/OperatorUtils.cpp:380-422); two index vectors are composed by the
library (vx355_compose_indices). torch only allocates the buffers: no torch
kernel runs inside the query.
"""
import ctypes as C

from . import abi

Q3_DATE = 9204  # 1995-03-15 in days since epoch


class DevBatch:
    """vx355_batch over device columns with an explicit row count."""

    def __init__(self, cols, num_rows):
        self.cols = cols
        self.num_rows = num_rows
        self._descs = (abi.Column * max(1, len(cols)))(*[c.descriptor() for c in cols])
        self.batch = abi.Batch(num_rows, len(cols), self._descs)

    def ref(self):
        return C.byref(self.batch)


def run_q3(ops, torch, tables, date=Q3_DATE, fuse_filters=True):
    """fuse_filters: the date filters of orders and lineitem run inside the probes
    (vx355_join_probe_set_input_filter) instead of as FilterProject passes in front of them."""
    if fuse_filters:
        return run_q3_fused(ops, torch, tables, date)
    return run_q3_unfused(ops, torch, tables, date)


def run_q3_fused(ops, torch, tables, date=Q3_DATE):
    """customer -> FilterProject -> HashBuild as below; orders and lineitem are probed straight from
    their flat columns with the date filter fused into the probe: the probes' mappings number table
    rows, so pass-through columns are dictionaries over ONE index vector and nothing is composed."""
    def flat(kind, t):
        return ops.DeviceColumn.from_ptr(kind, t.data_ptr(), int(t.shape[0]))

    def wrapped(kind, t, idx, n):
        return ops.DeviceColumn.from_ptr(kind, t.data_ptr(), n, None, abi.DICTIONARY, idx.data_ptr(),
                                         int(t.shape[0]))

    dev = tables["c_custkey"].device
    info = {}
    # -- customer (15 M rows, one 16-byte string column: the stand-alone FilterProject is 0.1 ms)
    nc = int(tables["c_custkey"].shape[0])
    cust = DevBatch([flat(abi.BIGINT, tables["c_custkey"]), flat(abi.VARCHAR, tables["c_mktsegment"])], nc)
    idx_c = torch.empty(max(1, nc), dtype=torch.int32, device=dev)
    mc = ops.filter_project_device(cust, [(1, abi.CMP_EQ, b"BUILDING")], [], idx_c.data_ptr(), [])
    b1 = ops.HashBuild([0], [abi.BIGINT], [], [], abi.JOIN_INNER)
    b1.add_input(DevBatch([wrapped(abi.BIGINT, tables["c_custkey"], idx_c, mc)], mc))
    t1 = b1.finish()
    info["customers_selected"] = mc

    # -- orders: probe(o_custkey) WHERE o_orderdate < date; every output page feeds the next build
    no = int(tables["o_orderkey"].shape[0])
    p1 = ops.HashProbe(t1, [0], abi.JOIN_INNER)
    p1.set_input_filter([(1, abi.CMP_LT, date)])
    p1.add_input(DevBatch([flat(abi.BIGINT, tables["o_custkey"]), flat(abi.INTEGER, tables["o_orderdate"])], no))
    b2 = ops.HashBuild([0], [abi.BIGINT], [1, 2], [abi.INTEGER, abi.INTEGER], abi.JOIN_INNER)
    cap1 = max(1, no // 4)   # TPC-H: 10 % of the orders join (48 % pass the date, 20 % of the customers are BUILDING)
    n1, keep = 0, []
    while True:
        ord_idx = torch.empty(cap1, dtype=torch.int32, device=dev)
        got, fin = p1.get_output_device(cap1, ord_idx.data_ptr(), None, None, [])
        if got:
            b2.add_input(DevBatch([wrapped(abi.BIGINT, tables["o_orderkey"], ord_idx, got),
                                   wrapped(abi.INTEGER, tables["o_orderdate"], ord_idx, got),
                                   wrapped(abi.INTEGER, tables["o_shippriority"], ord_idx, got)], got))
            keep.append(ord_idx)
        n1 += got
        if fin:
            break
    info["orders_selected"], info["orders_joined"] = no, n1   # rows the probe sees / emits
    t2 = b2.finish()

    # -- lineitem: probe(l_orderkey) WHERE l_shipdate > date; every output page is a batch of the aggregation
    nl = int(tables["l_orderkey"].shape[0])
    p2 = ops.HashProbe(t2, [0], abi.JOIN_INNER)
    p2.set_input_filter([(1, abi.CMP_GT, date)])
    p2.add_input(DevBatch([flat(abi.BIGINT, tables["l_orderkey"]), flat(abi.INTEGER, tables["l_shipdate"])], nl))
    agg = ops.HashAggregation([0, 1, 2], [abi.BIGINT, abi.INTEGER, abi.INTEGER],
                              [(abi.AGG_SUM, ops.PROJ(0), abi.DOUBLE)])
    agg.set_fused_input([], [[(3, 1.0, 0.0), (4, -1.0, 1.0)]])
    cap = max(1, nl // 16)   # TPC-H: 0.5 % of the lineitems join
    n2 = 0
    while True:
        li_idx = torch.empty(cap, dtype=torch.int32, device=dev)
        odate = torch.empty(cap, dtype=torch.int32, device=dev)
        oprio = torch.empty(cap, dtype=torch.int32, device=dev)
        nulls = torch.empty((cap // 64 + 1) * 2, dtype=torch.int64, device=dev)
        descs = (abi.OutColumn * 2)()
        for i, t in enumerate((odate, oprio)):
            descs[i].type_kind, descs[i].mem = abi.INTEGER, abi.MEM_DEVICE
            descs[i].values, descs[i].nulls = t.data_ptr(), nulls[i * (cap // 64 + 1):].data_ptr()
        got, fin = p2.get_output_device(cap, li_idx.data_ptr(), None, descs, [0, 1])
        if got:
            agg.add_input(DevBatch([wrapped(abi.BIGINT, tables["l_orderkey"], li_idx, got),
                                    ops.DeviceColumn.from_ptr(abi.INTEGER, odate.data_ptr(), got),
                                    ops.DeviceColumn.from_ptr(abi.INTEGER, oprio.data_ptr(), got),
                                    wrapped(abi.DOUBLE, tables["l_extendedprice"], li_idx, got),
                                    wrapped(abi.DOUBLE, tables["l_discount"], li_idx, got)], got))
        n2 += got
        if fin:
            break
    info["lineitems_selected"], info["lineitems_joined"] = nl, n2
    return _q3_drain(ops, torch, agg, dev, info)


def _q3_drain(ops, torch, agg, dev, info):
    """noMoreInput + the aggregation's groups as one device page: (l_orderkey, o_orderdate, o_shippriority, revenue)."""
    agg.no_more_input()
    groups = int(agg.stats().num_groups)
    outcap = max(1, groups)
    out = [torch.empty(outcap, dtype=torch.int64, device=dev), torch.empty(outcap, dtype=torch.int32, device=dev),
           torch.empty(outcap, dtype=torch.int32, device=dev), torch.empty(outcap, dtype=torch.float64, device=dev)]
    onulls = torch.empty((outcap // 64 + 1) * 4, dtype=torch.int64, device=dev)
    od = (abi.OutColumn * 4)()
    for i, (t, k) in enumerate(zip(out, (abi.BIGINT, abi.INTEGER, abi.INTEGER, abi.DOUBLE))):
        od[i].type_kind, od[i].mem = k, abi.MEM_DEVICE
        od[i].values, od[i].nulls = t.data_ptr(), onulls[i * (outcap // 64 + 1):].data_ptr()
    total = 0
    n, fin = C.c_int32(), C.c_int32(0)
    if groups:
        ops._check(ops.lib().vx355_agg_get_output(agg.h, od, 4, outcap, C.byref(n), C.byref(fin)))
        total = n.value
        assert fin.value
    info["groups"] = total
    info["agg_mode"] = int(agg.stats().hash_mode)
    return [t[:total] for t in out], info


def run_q3_unfused(ops, torch, tables, date=Q3_DATE):
    """tables: dict of torch tensors in HBM:
      c_custkey int64, c_mktsegment int32[n,4] (16-byte StringViews),
      o_orderkey int64, o_custkey int64, o_orderdate int32, o_shippriority int32,
      l_orderkey int64, l_shipdate int32, l_extendedprice float64, l_discount float64.
    Returns (keys..., revenue) tensors of the aggregation output plus stage sizes."""
    def flat(kind, t):
        return ops.DeviceColumn.from_ptr(kind, t.data_ptr(), int(t.shape[0]))

    def wrapped(kind, t, idx, n):
        return ops.DeviceColumn.from_ptr(kind, t.data_ptr(), n, None, abi.DICTIONARY, idx.data_ptr(),
                                         int(t.shape[0]))

    dev = tables["c_custkey"].device
    info = {}

    def select(batch, terms, n):
        idx = torch.empty(max(1, n), dtype=torch.int32, device=dev)
        m = ops.filter_project_device(batch, terms, [], idx.data_ptr(), [])
        return idx, m

    # -- customer
    nc = int(tables["c_custkey"].shape[0])
    cust = DevBatch([flat(abi.BIGINT, tables["c_custkey"]), flat(abi.VARCHAR, tables["c_mktsegment"])], nc)
    idx_c, mc = select(cust, [(1, abi.CMP_EQ, b"BUILDING")], nc)
    b1 = ops.HashBuild([0], [abi.BIGINT], [], [], abi.JOIN_INNER)
    b1.add_input(DevBatch([wrapped(abi.BIGINT, tables["c_custkey"], idx_c, mc)], mc))
    t1 = b1.finish()
    info["customers_selected"] = mc

    # -- orders
    no = int(tables["o_orderkey"].shape[0])
    orders = DevBatch([flat(abi.INTEGER, tables["o_orderdate"])], no)
    idx_o, mo = select(orders, [(0, abi.CMP_LT, date)], no)
    p1 = ops.HashProbe(t1, [0], abi.JOIN_INNER)
    p1.add_input(DevBatch([wrapped(abi.BIGINT, tables["o_custkey"], idx_o, mo)], mo))
    map_o = torch.empty(max(1, mo), dtype=torch.int32, device=dev)
    n1, fin = p1.get_output_device(max(1, mo), map_o.data_ptr(), None, None, [])
    assert fin
    # pass-through columns of the probe output = dictionary over FilterProject's dictionary: the
    # library composes the two index vectors (vx355_compose_indices, wrapChild of a wrapped vector)
    ord_idx = torch.empty(max(1, n1), dtype=torch.int32, device=dev)
    if n1:
        ops.compose_indices_device(idx_o.data_ptr(), mo, map_o.data_ptr(), n1, ord_idx.data_ptr())
    info["orders_selected"], info["orders_joined"] = mo, n1
    b2 = ops.HashBuild([0], [abi.BIGINT], [1, 2], [abi.INTEGER, abi.INTEGER], abi.JOIN_INNER)
    if n1:
        b2.add_input(DevBatch([wrapped(abi.BIGINT, tables["o_orderkey"], ord_idx, n1),
                               wrapped(abi.INTEGER, tables["o_orderdate"], ord_idx, n1),
                               wrapped(abi.INTEGER, tables["o_shippriority"], ord_idx, n1)], n1))
    t2 = b2.finish()

    # -- lineitem
    nl = int(tables["l_orderkey"].shape[0])
    line = DevBatch([flat(abi.INTEGER, tables["l_shipdate"])], nl)
    idx_l, ml = select(line, [(0, abi.CMP_GT, date)], nl)
    p2 = ops.HashProbe(t2, [0], abi.JOIN_INNER)
    p2.add_input(DevBatch([wrapped(abi.BIGINT, tables["l_orderkey"], idx_l, ml)], ml))
    cap = max(1, ml)
    map_l = torch.empty(cap, dtype=torch.int32, device=dev)
    odate = torch.empty(cap, dtype=torch.int32, device=dev)
    oprio = torch.empty(cap, dtype=torch.int32, device=dev)
    nulls = torch.empty((cap // 64 + 1) * 2, dtype=torch.int64, device=dev)
    descs = (abi.OutColumn * 2)()
    for i, t in enumerate((odate, oprio)):
        descs[i].type_kind, descs[i].mem = abi.INTEGER, abi.MEM_DEVICE
        descs[i].values, descs[i].nulls = t.data_ptr(), nulls[i * (cap // 64 + 1):].data_ptr()
    n2, fin = p2.get_output_device(cap, map_l.data_ptr(), None, descs, [0, 1])
    assert fin
    info["lineitems_selected"], info["lineitems_joined"] = ml, n2
    li_idx = torch.empty(max(1, n2), dtype=torch.int32, device=dev)
    if n2:
        ops.compose_indices_device(idx_l.data_ptr(), ml, map_l.data_ptr(), n2, li_idx.data_ptr())

    # -- aggregation: group by (l_orderkey, o_orderdate, o_shippriority), sum(ep * (1 - disc))
    agg = ops.HashAggregation([0, 1, 2], [abi.BIGINT, abi.INTEGER, abi.INTEGER],
                              [(abi.AGG_SUM, ops.PROJ(0), abi.DOUBLE)])
    agg.set_fused_input([], [[(3, 1.0, 0.0), (4, -1.0, 1.0)]])
    if n2:
        agg.add_input(DevBatch([wrapped(abi.BIGINT, tables["l_orderkey"], li_idx, n2),
                                ops.DeviceColumn.from_ptr(abi.INTEGER, odate.data_ptr(), n2),
                                ops.DeviceColumn.from_ptr(abi.INTEGER, oprio.data_ptr(), n2),
                                wrapped(abi.DOUBLE, tables["l_extendedprice"], li_idx, n2),
                                wrapped(abi.DOUBLE, tables["l_discount"], li_idx, n2)], n2))
    agg.no_more_input()
    groups = int(agg.stats().num_groups)
    outcap = max(1, groups)
    out = [torch.empty(outcap, dtype=torch.int64, device=dev), torch.empty(outcap, dtype=torch.int32, device=dev),
           torch.empty(outcap, dtype=torch.int32, device=dev), torch.empty(outcap, dtype=torch.float64, device=dev)]
    onulls = torch.empty((outcap // 64 + 1) * 4, dtype=torch.int64, device=dev)
    od = (abi.OutColumn * 4)()
    for i, (t, k) in enumerate(zip(out, (abi.BIGINT, abi.INTEGER, abi.INTEGER, abi.DOUBLE))):
        od[i].type_kind, od[i].mem = k, abi.MEM_DEVICE
        od[i].values, od[i].nulls = t.data_ptr(), onulls[i * (outcap // 64 + 1):].data_ptr()
    total = 0
    n, fin = C.c_int32(), C.c_int32(0)
    if groups:
        ops._check(ops.lib().vx355_agg_get_output(agg.h, od, 4, outcap, C.byref(n), C.byref(fin)))
        total = n.value
        assert fin.value
    info["groups"] = total
    info["agg_mode"] = int(agg.stats().hash_mode)
    return [t[:total] for t in out], info
