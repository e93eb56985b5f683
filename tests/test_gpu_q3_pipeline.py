"""TPC-H Q3 end to end on the GPU (velox_amd/tpch.py: two hash joins and a
three-key aggregation chained through dictionary-wrapped columns) against
pandas on a small TPC-H-shaped data set."""
import numpy as np
import pandas as pd
import pytest

from velox_amd import abi, tpch

pytestmark = pytest.mark.gpu


def _tables(rng, n_cust=3000, n_orders=30000):
    segs = [b"AUTOMOBILE", b"BUILDING", b"FURNITURE", b"MACHINERY", b"HOUSEHOLD"]
    c_custkey = np.arange(1, n_cust + 1, dtype=np.int64)
    c_seg = [segs[i] for i in rng.integers(0, 5, n_cust)]
    seq = np.arange(n_orders, dtype=np.int64)
    o_orderkey = (seq // 8) * 32 + (seq % 8)
    valid_cust = c_custkey[c_custkey % 3 != 0]
    o_custkey = valid_cust[rng.integers(0, len(valid_cust), n_orders)]
    o_orderdate = rng.integers(8035, 10440, n_orders).astype(np.int32)
    o_shippriority = np.zeros(n_orders, dtype=np.int32)
    counts = rng.integers(1, 8, n_orders)
    li = np.repeat(seq, counts)
    l_orderkey = o_orderkey[li]
    l_shipdate = (o_orderdate[li] + rng.integers(1, 122, len(li))).astype(np.int32)
    l_ep = rng.integers(90000, 10500000, len(li)) / 128.0      # dyadic: sums are exact
    l_disc = rng.integers(0, 11, len(li)) / 64.0
    return dict(c_custkey=c_custkey, c_mktsegment=c_seg, o_orderkey=o_orderkey, o_custkey=o_custkey,
                o_orderdate=o_orderdate, o_shippriority=o_shippriority, l_orderkey=l_orderkey,
                l_shipdate=l_shipdate, l_extendedprice=l_ep, l_discount=l_disc)


@pytest.mark.parametrize("fuse_filters", [True, False])
def test_q3_pipeline_matches_pandas(vx, fuse_filters):
    """fuse_filters: the date filters inside the probes (vx355_join_probe_set_input_filter; the probes'
    output pages are smaller than their outputs here, so the paged hand-over to the next build / the
    aggregation runs too) or as FilterProject passes in front of them."""
    import torch
    rng = np.random.default_rng(77)
    t = _tables(rng)
    dev = torch.device("cuda", 0)
    views, _keep = abi.string_views(t["c_mktsegment"])
    tables = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in t.items() if k != "c_mktsegment"}
    tables["c_mktsegment"] = torch.from_numpy(np.ascontiguousarray(views).view(np.int32).reshape(-1, 4)).to(dev)
    out, info = tpch.run_q3(vx, torch, tables, fuse_filters=fuse_filters)
    cust = pd.DataFrame({"c_custkey": t["c_custkey"], "seg": [s.decode() for s in t["c_mktsegment"]]})
    orders = pd.DataFrame({k: t[k] for k in ("o_orderkey", "o_custkey", "o_orderdate", "o_shippriority")})
    line = pd.DataFrame({k: t[k] for k in ("l_orderkey", "l_shipdate", "l_extendedprice", "l_discount")})
    j1 = orders[orders.o_orderdate < tpch.Q3_DATE].merge(cust[cust.seg == "BUILDING"], left_on="o_custkey",
                                                          right_on="c_custkey")
    j2 = line[line.l_shipdate > tpch.Q3_DATE].merge(j1, left_on="l_orderkey", right_on="o_orderkey")
    j2["rev"] = j2.l_extendedprice * (1 - j2.l_discount)
    want = j2.groupby(["l_orderkey", "o_orderdate", "o_shippriority"])["rev"].sum()
    assert info["orders_joined"] == len(j1) and info["lineitems_joined"] == len(j2)
    got = pd.Series(out[3].cpu().numpy(),
                    index=pd.MultiIndex.from_arrays([out[0].cpu().numpy(), out[1].cpu().numpy(), out[2].cpu().numpy()]))
    assert len(got) == len(want) and len(got) > 50
    got = got.sort_index()
    want = want.sort_index()
    assert (got.index.to_frame().values == want.index.to_frame().values).all()
    assert (got.values == want.values).all()   # dyadic inputs: bit-identical revenue sums
