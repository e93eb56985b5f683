"""Parity at BASELINE.json's full sizes, through properties that do not need the CPU
oracle to chew through 10^9 rows: integer results are checked bit-exactly against torch
integer reductions of the same HBM-resident columns, DOUBLE results against exact
integer sums where the data allows it, and the partial -> final split, join
completeness and pair validity against the algebra of the operators. The inputs are
bench.py's generators (the workloads BENCH lines are quoted on)."""
import numpy as np
import pytest

from velox_amd import abi
from gpu_util import ulp_distance

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu(vx):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def test_q1_sf100_counts_and_integer_sums_are_exact_and_partial_final_agrees(vx, torch_gpu):
    """TPC-H Q1 at SF100 (600 037 902 rows), fused FilterProject + HashAggregation."""
    torch = torch_gpu
    import bench
    from velox_amd import dist as vdist
    n = 600_037_902
    wl = bench.Q1(torch, n, "cuda:0", 4321)
    out = wl.step()
    c = wl.c
    keep = c["ship"] <= bench.Q1_CUTOFF
    code = (c["rf"][:, 1].to(torch.int64) * 256 + c["ls"][:, 1].to(torch.int64))[keep]
    groups, inverse, counts = torch.unique(code, return_inverse=True, return_counts=True)
    qty_sum = torch.zeros(len(groups), dtype=torch.int64, device=code.device).index_add_(
        0, inverse, c["qty"][keep].to(torch.int64))
    exp = {int(g): (int(cnt), int(q)) for g, cnt, q in zip(groups.tolist(), counts.tolist(), qty_sum.tolist())}
    del code, inverse, keep
    rf, ls = out[0][0], out[1][0]
    assert len(rf) == len(exp) and int(sum(out[9][0])) == sum(v[0] for v in exp.values())
    for i in range(len(rf)):
        cnt, q = exp[rf[i][0] * 256 + ls[i][0]]
        assert int(out[9][0][i]) == cnt                      # count(*): bit exact
        assert out[2][0][i] == float(q)                      # sum(l_quantity): integers as doubles, exact
        assert out[6][0][i] == float(q) / cnt                # avg = sum / count (AverageAggregateBase.h)
    # partial over two halves + final == single pass: keys, counts, integer sums exact, other sums <= 1 ULP
    half = (n // 2) & ~63
    parts = []
    for lo, hi in ((0, half), (half, n)):
        scan = bench.DevBatch([bench.dcol(abi.VARCHAR, c["rf"][lo:hi]), bench.dcol(abi.VARCHAR, c["ls"][lo:hi]),
                               bench.dcol(abi.DOUBLE, c["qty"][lo:hi]), bench.dcol(abi.DOUBLE, c["ep"][lo:hi]),
                               bench.dcol(abi.DOUBLE, c["disc"][lo:hi]), bench.dcol(abi.DOUBLE, c["tax"][lo:hi]),
                               bench.dcol(abi.INTEGER, c["ship"][lo:hi])], hi - lo)
        op = vx.HashAggregation(bench.Q1_KEYS[0], bench.Q1_KEYS[1], bench.Q1.FUSED_AGGS, abi.STEP_PARTIAL)
        op.set_fused_input(bench.Q1_TERMS, bench.Q1_PROJ)
        op.add_input(scan)
        op.no_more_input()
        parts.append(vx.collect_output(op, 1024))
    kinds = vdist.partial_kinds(bench.Q1_KEYS[1], bench.Q1.FUSED_AGGS)
    mat = np.concatenate([vdist.encode_partial(p, kinds, len(p[0][1])) for p in parts])
    fin = vx.HashAggregation([0, 1], bench.Q1_KEYS[1], vdist.final_aggs_for(bench.Q1.FUSED_AGGS, 2), abi.STEP_FINAL)
    fin.add_input(vdist.decode_partials(mat, kinds))
    fin.no_more_input()
    merged = vx.collect_output(fin, 1024)
    order = {(rf[i], ls[i]): i for i in range(len(rf))}
    for j in range(len(merged[0][0])):
        i = order[(merged[0][0][j], merged[1][0][j])]
        assert merged[9][0][j] == out[9][0][i] and merged[2][0][j] == out[2][0][i]
        for col in (3, 4, 5, 6, 7, 8):
            assert ulp_distance(np.array([merged[col][0][j]]), np.array([out[col][0][i]]))[0] <= 1, col


def exact_group_sums_torch(torch, v, group, num_groups):
    """Exact per-group sums of a float64 tensor, as Python Fractions, computed on the GPU in
    integer arithmetic: every finite double is M * 2^(E-1075) with a 53-bit integer M; M is
    split into a 27-bit high and a 26-bit low part which are summed per (group, exponent)
    bucket in int64 (no bucket can overflow below 2^36 rows), and the buckets are combined
    with Python's unbounded integers."""
    from fractions import Fraction
    bits = v.view(torch.int64)
    e = (bits >> 52) & 0x7FF
    frac = bits & ((1 << 52) - 1)
    assert int(e.max()) < 0x7FF, "inf / NaN in the input"
    m = torch.where(e > 0, frac | (1 << 52), frac)
    e = torch.where(e > 0, e, torch.ones_like(e))          # subnormals: 2^(1-1075)
    m = torch.where(bits < 0, -m, m)
    del bits, frac
    hi = m >> 26                                            # floor: m = hi * 2^26 + lo, 0 <= lo < 2^26
    lo = m - (hi << 26)
    del m
    idx = group * 2048 + e
    del e
    sum_hi = torch.zeros(num_groups * 2048, dtype=torch.int64, device=v.device).index_add_(0, idx, hi)
    sum_lo = torch.zeros(num_groups * 2048, dtype=torch.int64, device=v.device).index_add_(0, idx, lo)
    del idx, hi, lo
    sum_hi, sum_lo = sum_hi.cpu().tolist(), sum_lo.cpu().tolist()
    out = []
    for g in range(num_groups):
        total = Fraction(0)
        for ex in range(2048):
            h, l = sum_hi[g * 2048 + ex], sum_lo[g * 2048 + ex]
            if h or l:
                total += Fraction((h << 26) + l) * Fraction(2) ** (ex - 1075)
        out.append(total)
    return out


def test_exact_group_sums_torch_agrees_with_fsum(torch_gpu):
    import math
    torch = torch_gpu
    g = torch.Generator(device="cuda:0")
    g.manual_seed(5)
    n = 200000
    v = (torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.3) * \
        (10.0 ** torch.randint(-30, 30, (n,), device="cuda:0", generator=g).to(torch.float64))
    v[:7] = torch.tensor([0.0, -0.0, 5e-324, -2.5e-320, 1.7e308, -1.7e308, 1e-310], dtype=torch.float64)
    grp = torch.randint(0, 3, (n,), dtype=torch.int64, device="cuda:0", generator=g)
    got = exact_group_sums_torch(torch, v, grp, 3)
    hv, hg = v.cpu().numpy(), grp.cpu().numpy()
    for i in range(3):
        assert float(got[i]) == math.fsum(hv[hg == i].tolist())


def test_q1_sf100_double_sums_within_one_ulp_of_exact_integer_reference(vx, torch_gpu):
    """The configuration BENCH is quoted on: all five DOUBLE sums of TPC-H Q1 at SF100
    (150 M rows per group) against an EXACT reference that does not come from this library:
    torch computes the projections element-wise in IEEE double (the same roundings the fused
    FilterProject makes, exec/FilterProject.cpp:102-275), the per-group sums of those doubles
    are then taken exactly in integer arithmetic (exact_group_sums_torch). Sums must be within
    1 ULP of the correctly rounded exact sum (north_star), averages within 2 (one division on
    top)."""
    torch = torch_gpu
    import bench
    n = 600_037_902
    wl = bench.Q1(torch, n, "cuda:0", 4321)
    out = wl.step()
    c = wl.c
    keep = c["ship"] <= bench.Q1_CUTOFF
    code = (c["rf"][:, 1].to(torch.int64) * 256 + c["ls"][:, 1].to(torch.int64))[keep]
    groups, inverse, counts = torch.unique(code, return_inverse=True, return_counts=True)
    del code
    ng = len(groups)
    order = {int(g): i for i, g in enumerate(groups.tolist())}
    rf, ls = out[0][0], out[1][0]
    assert len(rf) == ng
    row_of = [order[rf[i][0] * 256 + ls[i][0]] for i in range(ng)]
    cnt = counts.tolist()
    ep = c["ep"][keep]
    disc = c["disc"][keep]
    one_minus = disc * -1.0 + 1.0             # vx355_factor: scale * column + offset
    disc_price = (ep * 1.0 + 0.0) * one_minus
    charge = disc_price * (c["tax"][keep] * 1.0 + 1.0)
    del one_minus
    # output columns: 2 sum(qty) 3 sum(ep) 4 sum(disc_price) 5 sum(charge) 6 avg(qty) 7 avg(ep) 8 avg(disc) 9 count
    worst = {}
    for name, tensor, sum_col, avg_col in (("qty", c["qty"][keep], 2, 6), ("ep", ep, 3, 7),
                                            ("disc_price", disc_price, 4, None), ("charge", charge, 5, None),
                                            ("disc", disc, None, 8)):
        exact = exact_group_sums_torch(torch, tensor, inverse, ng)
        for i in range(ng):
            ex = exact[row_of[i]]
            if sum_col is not None:
                d = ulp_distance(np.array([out[sum_col][0][i]]), np.array([float(ex)]))[0]
                worst[name] = max(worst.get(name, 0), d)
                assert d <= 1, (name, i, out[sum_col][0][i], float(ex))
            if avg_col is not None:
                d = ulp_distance(np.array([out[avg_col][0][i]]), np.array([float(ex / cnt[row_of[i]])]))[0]
                worst["avg_" + name] = max(worst.get("avg_" + name, 0), d)
                assert d <= 2, ("avg " + name, i)
            assert int(out[9][0][i]) == cnt[row_of[i]]
    print("Q1 SF100 worst ULP distance from the exact sums:", worst)


def test_q1_sf100_four_keys_six_aggregates_on_the_specialised_kernel(vx, torch_gpu):
    """BASELINE.json's wording of configs[1] at full size (bench.py --workload q1x4): 600 037 902 rows,
    8 scan columns, 4 grouping keys (196 groups), 6 aggregates, fused filter. Counts and sum(qty)
    bit exact against torch integer reductions, the other DOUBLE sums within 1 ULP of the exact
    integer reference, avg = sum / count, groups in first-seen order, and the whole batch on
    k_agg_fast (ahead-of-time instance: no hiprtc, no interpreting kernel)."""
    torch = torch_gpu
    import bench
    n = 600_037_902
    wl = bench.Q1FourKeys(torch, n, "cuda:0", 4321)
    vx.profile_reset()
    vx.profile_enable(True)
    op = wl.operator(abi.STEP_SINGLE)
    op.add_input(wl.scan)
    op.no_more_input()
    out = vx.collect_output(op, 1024)
    vx.profile_enable(False)
    names = vx.profile()
    assert "k_agg_fast" in names and "k_agg_lds" not in names and "k_agg_global" not in names, names
    c = wl.c
    keep = c["ship"] <= bench.Q1_CUTOFF
    code_all = ((c["rf"][:, 1].to(torch.int64) * 256 + c["ls"][:, 1].to(torch.int64)) * 16 +
                c["lnum"].to(torch.int64)) * 16 + c["mode"].to(torch.int64)
    code = code_all[keep]
    groups, inverse, counts = torch.unique(code, return_inverse=True, return_counts=True)
    ng = len(groups)
    assert ng == 4 * 7 * 7
    order = {int(g): i for i, g in enumerate(groups.tolist())}
    rf, ls, lnum, mode = out[0][0], out[1][0], out[2][0], out[3][0]
    assert len(rf) == ng
    got_code = [((rf[i][0] * 256 + ls[i][0]) * 16 + int(lnum[i])) * 16 + int(mode[i]) for i in range(ng)]
    row_of = [order[g] for g in got_code]
    assert sorted(row_of) == list(range(ng))
    # first-seen order (GroupingSet.cpp:828-839): the first passing row of every group, ascending
    first_row = torch.full((ng,), n, dtype=torch.int64, device=code.device).scatter_reduce_(
        0, inverse, torch.nonzero(keep).flatten(), reduce="amin")
    assert [int(x) for x in torch.argsort(first_row).tolist()] == row_of
    cnt = counts.tolist()
    qty_sum = torch.zeros(ng, dtype=torch.int64, device=code.device).index_add_(
        0, inverse, c["qty"][keep].to(torch.int64)).tolist()
    # output columns: 4 sum(qty) 5 sum(ep) 6 sum(disc_price) 7 avg(qty) 8 avg(disc) 9 count(*)
    for i in range(ng):
        g = row_of[i]
        assert int(out[9][0][i]) == cnt[g]
        assert out[4][0][i] == float(qty_sum[g])
        assert out[7][0][i] == float(qty_sum[g]) / cnt[g]
    ep = c["ep"][keep]
    disc = c["disc"][keep]
    disc_price = (ep * 1.0 + 0.0) * (disc * -1.0 + 1.0)
    worst = {}
    for name, tensor, sum_col, avg_col in (("ep", ep, 5, None), ("disc_price", disc_price, 6, None),
                                            ("disc", disc, None, 8)):
        exact = exact_group_sums_torch(torch, tensor, inverse, ng)
        for i in range(ng):
            ex = exact[row_of[i]]
            if sum_col is not None:
                d = ulp_distance(np.array([out[sum_col][0][i]]), np.array([float(ex)]))[0]
                worst[name] = max(worst.get(name, 0), d)
                assert d <= 1, (name, i, out[sum_col][0][i], float(ex))
            else:
                d = ulp_distance(np.array([out[avg_col][0][i]]), np.array([float(ex / cnt[row_of[i]])]))[0]
                worst["avg_" + name] = max(worst.get("avg_" + name, 0), d)
                assert d <= 2, ("avg " + name, i)
    print("Q1 4-key SF100 worst ULP distance from the exact sums:", worst)


def test_q3_sf100_join_is_complete_and_every_pair_is_valid(vx, torch_gpu):
    """The dominant join of TPC-H Q3 at SF100: ~14.6 M build rows, ~323 M probe rows."""
    torch = torch_gpu
    import bench
    wl = bench.Q3(torch, 600_037_902, "cuda:0", 999)
    n = wl.step()
    # completeness: orders keys are unique, so the inner join emits one pair per probe key present in the build side
    sorted_keys, order = torch.sort(wl.bkey)
    pos = torch.searchsorted(sorted_keys, wl.pkey).clamp_(max=len(sorted_keys) - 1)
    present = sorted_keys[pos] == wl.pkey
    assert n == int(present.sum())
    mapping, brows = wl.mapping[:n].to(torch.int64), wl.brows[:n].to(torch.int64)
    # ascending probe rows, exactly the probe rows with a match
    assert bool((mapping[1:] > mapping[:-1]).all())
    assert bool((mapping == torch.nonzero(present).flatten()).all())
    # every pair joins equal keys; the gathered payload is the build row's payload
    assert bool((wl.bkey[brows] == wl.pkey[mapping]).all())
    assert bool((wl.odate_out[:n] == wl.bdate[brows]).all())
    # semi + anti partition the probe side
    totals = {}
    for jt in (abi.JOIN_LEFT_SEMI_FILTER, abi.JOIN_ANTI):
        b = vx.HashBuild([0], [abi.BIGINT], [], [], jt)
        b.add_input(wl.build)
        p = vx.HashProbe(b.finish(), [0], jt)
        p.add_input(wl.probe)
        totals[jt], fin = p.get_output_device(wl.probe_rows, wl.mapping.data_ptr(), 0)
        assert fin
    assert totals[abi.JOIN_LEFT_SEMI_FILTER] == n and totals[abi.JOIN_ANTI] == wl.probe_rows - n


@pytest.mark.parametrize("unordered", [False, True])
@pytest.mark.parametrize("sparse", [False, True])
def test_c4_one_billion_rows_counts_exact_sums_close(vx, torch_gpu, sparse, unordered):
    """BASELINE config 4 at full size (dense keys: radix-partitioned LDS path; sparse keys:
    open addressing on the value): sum(v) + count(*) per group against torch reductions. unordered:
    VX355_AGG_UNORDERED_OUTPUT, i.e. the compact records of round 5 (12 bytes over the direct-index table,
    16-byte {key, operand} over the open-addressing one) through both scatter levels and the folds at 10^9 rows."""
    torch = torch_gpu
    import bench
    n, distinct = 1_000_000_000, 100_000_000
    g = torch.Generator(device="cuda:0")
    g.manual_seed(7)
    j = torch.randint(0, distinct, (n,), dtype=torch.int64, device="cuda:0", generator=g)
    v = torch.randint(0, 1 << 20, (n,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.float64) / 1024.0
    k = j
    if sparse:
        k = (j * -7046029254386353131) ^ 0x5DEECE66D   # odd multiplier mod 2^64: a bijection, keys all over int64
    torch.cuda.synchronize()   # the library runs on its own stream: device inputs must be complete
    op = vx.HashAggregation([0], [abi.BIGINT], [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)],
                            abi.STEP_SINGLE, flags=abi.AGG_UNORDERED_OUTPUT if unordered else 0)
    op.add_input(bench.DevBatch([bench.dcol(abi.BIGINT, k), bench.dcol(abi.DOUBLE, v)], n))
    op.no_more_input()
    st = op.stats()
    assert st.input_rows == n
    assert st.compact_record_launches == (st.radix_launches if unordered else 0) and st.radix_launches >= 1
    if not sparse:
        assert st.hash_mode == abi.MODE_ARRAY and st.radix_launches >= 1   # 10^9 rows are ONE radix chunk since round 3
    else:
        assert st.hash_mode == abi.MODE_NORMALIZED_KEY
    cap = 1 << 24
    keys = torch.empty(cap, dtype=torch.int64, device="cuda:0")
    sums = torch.empty(cap, dtype=torch.float64, device="cuda:0")
    cnts = torch.empty(cap, dtype=torch.int64, device="cuda:0")
    nulls = [torch.empty(cap // 64 + 1, dtype=torch.int64, device="cuda:0") for _ in range(3)]
    descs = (abi.OutColumn * 3)()
    for i, (kind, t) in enumerate(((abi.BIGINT, keys), (abi.DOUBLE, sums), (abi.BIGINT, cnts))):
        descs[i].type_kind, descs[i].mem = kind, abi.MEM_DEVICE
        descs[i].values, descs[i].nulls = t.data_ptr(), nulls[i].data_ptr()
    exp_cnt = torch.bincount(j, minlength=distinct)
    exp_sum = torch.zeros(distinct, dtype=torch.int64, device="cuda:0").index_add_(0, j, (v * 1024.0).to(torch.int64))
    import ctypes as C
    seen, total_groups = torch.zeros(distinct, dtype=torch.bool, device="cuda:0"), 0
    while True:
        m, fin = C.c_int32(), C.c_int32()
        vx._check(vx.lib().vx355_agg_get_output(op.h, descs, 3, cap, C.byref(m), C.byref(fin)))
        m = m.value
        if m:
            kk = keys[:m]
            jj = ((kk ^ 0x5DEECE66D) * -1018231460777725123) if sparse else kk   # inverse of the bijection
            assert bool(((jj >= 0) & (jj < distinct)).all()) and not bool(seen[jj].any())
            seen[jj] = True
            assert bool((cnts[:m] == exp_cnt[jj]).all())                          # counts: bit exact
            # dyadic values: the exact sum is an integer multiple of 2^-10 below 2^53 -> bit exact too
            assert bool((sums[:m] == exp_sum[jj].to(torch.float64) / 1024.0).all())
            total_groups += m
        if fin.value:
            break
    assert total_groups == int((exp_cnt > 0).sum()) == st.num_groups
