"""sum(BIGINT) overflow is a function of the ROWS, not of the order in which the GPU adds them.

The reference checks its running sum in input order (checkedPlus, vector/AggregationHook.h:126-135):
it throws as soon as a prefix leaves int64. No parallel order sees the same prefixes, and a check on
whatever partial sums a kernel happens to form would raise - or not - at random on mixed-sign data.
libvx355 keeps a 128-bit total per group (low word + carries, both commutative) and checks it once,
when it is read out: "integer overflow" iff the exact total does not fit int64. The oracle exposes
both rules (oracle_lib.set_sum_overflow_rule); these tests pin the GPU to the TOTAL rule on every
kernel that adds integers - LDS (k_agg_lds), HBM atomics incl. the per-wave hot-key combining
(k_agg_global), the radix fold (k_rp_aggregate), the generic hash mode - and across partial -> final.
Where no prefix overflows the two rules agree, and the GPU equals the reference bit for bit."""
import itertools

import numpy as np
import pytest

from velox_amd import abi
from gpu_util import assert_columns_equal, batch_of, run_agg

pytestmark = pytest.mark.gpu

BIG = 1 << 62
AGGS = [(abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
KINDS = [abi.BIGINT, abi.BIGINT, abi.BIGINT]


@pytest.fixture
def total_rule(oracle):
    oracle.set_sum_overflow_rule(oracle.SUM_RULE_TOTAL)
    yield oracle
    oracle.set_sum_overflow_rule(oracle.SUM_RULE_REFERENCE)


def _data(rng, n, groups, special_group, special_values):
    """n filler rows of small mixed-sign values over 'groups' keys, plus the special values
    (a permutation of +-2^62 ...) planted at random rows of one group."""
    k = rng.integers(0, groups, n).astype(np.int64)
    v = rng.integers(-1000, 1000, n).astype(np.int64)
    at = rng.choice(n, len(special_values), replace=False)
    at.sort()
    k[at] = special_group
    v[at] = np.array(special_values, dtype=np.int64)
    return k, v


def _paths(monkeypatch):
    """name -> (environment, number of rows, number of groups, key transform, expected kernel)."""
    return {
        "lds": ({"VX355_AGG_COALESCE_ROWS": "0", "VX355_AGG_NO_FAST": "1"}, 200_000, 100, None, "k_agg_lds"),
        "global": ({"VX355_AGG_RADIX_MIN_ROWS": "-1", "VX355_AGG_COALESCE_ROWS": "0"}, 400_000, 30_000, None, "k_agg_global"),
        "global_hot": ({"VX355_AGG_RADIX_MIN_ROWS": "-1", "VX355_AGG_COALESCE_ROWS": "0"}, 400_000, 30_000, "hot", "k_agg_global"),
        "radix": ({"VX355_AGG_RADIX_MIN_ROWS": "100000", "VX355_AGG_COALESCE_ROWS": "0"}, 600_000, 30_000, None, "k_rp_aggregate"),
        "generic": ({"VX355_AGG_COALESCE_ROWS": "0"}, 200_000, 5_000, "double", "k_agg_generic"),
    }


@pytest.mark.parametrize("path", ["lds", "global", "global_hot", "radix", "generic"])
def test_transient_prefix_overflow_does_not_depend_on_the_order(total_rule, vx, monkeypatch, path):
    """[2^62, 2^62, -2^62, -2^62] in every order: the exact total is the filler's sum, no prefix rule
    can be followed in parallel, the GPU never raises and returns the exact total; and the reference
    rule would have raised for the orders whose prefix reaches 2^63."""
    oracle = total_rule
    env, n, groups, transform, kernel = _paths(monkeypatch)[path]
    for name, value in env.items():
        monkeypatch.setenv(name, value)
    rng = np.random.default_rng(17)
    raised_by_reference = 0
    for perm in sorted(set(itertools.permutations([BIG, BIG, -BIG, -BIG]))):
        k, v = _data(rng, n, groups, 7, perm)
        if transform == "hot":
            k[rng.random(n) < 0.3] = 7          # a third of the rows on the special key: wave combining
        key_types = [abi.BIGINT]
        cols = [k, v]
        if transform == "double":
            cols = [k.astype(np.float64) / 4, v]  # DOUBLE keys: generic hash mode
            key_types = [abi.DOUBLE]
        batch = batch_of(cols)
        vx.profile_reset()
        vx.profile_enable(True)
        got, _ = run_agg(vx, [batch], [0], key_types, AGGS, max_rows=1 << 20)
        vx.profile_enable(False)
        assert kernel in vx.profile(), (path, sorted(vx.profile()))
        exp, _ = run_agg(oracle, [batch], [0], key_types, AGGS, max_rows=1 << 20)
        assert_columns_equal(got, exp, [key_types[0], abi.BIGINT, abi.BIGINT], what=f"{path} {perm}")
        oracle.set_sum_overflow_rule(oracle.SUM_RULE_REFERENCE)
        try:
            run_agg(oracle, [batch], [0], key_types, AGGS, max_rows=1 << 20)
        except oracle.OracleError as e:
            assert "integer overflow" in str(e)
            raised_by_reference += 1
        finally:
            oracle.set_sum_overflow_rule(oracle.SUM_RULE_TOTAL)
    assert raised_by_reference >= 1   # (+,+,-,-) reaches 2^63 whatever the filler rows in between add


@pytest.mark.parametrize("path", ["lds", "global", "radix", "generic"])
def test_total_overflow_raises_on_every_kernel(total_rule, vx, monkeypatch, path):
    oracle = total_rule
    env, n, groups, transform, kernel = _paths(monkeypatch)[path]
    for name, value in env.items():
        monkeypatch.setenv(name, value)
    rng = np.random.default_rng(18)
    for special in ([BIG, BIG, BIG], [-BIG, -BIG, -BIG, -BIG], [BIG, -BIG, BIG, BIG, -1, BIG - 1_000_000]):
        k, v = _data(rng, n, groups, 3, special)
        v[k == 3] = np.where(np.isin(v[k == 3], special), v[k == 3], 0)   # the special group holds only the special values
        key_types = [abi.BIGINT]
        cols = [k, v]
        if transform == "double":
            cols, key_types = [k.astype(np.float64) / 4, v], [abi.DOUBLE]
        batch = batch_of(cols)
        total = int(sum(special))
        fits = -(1 << 63) <= total < (1 << 63)
        for impl in (oracle, vx):
            err = impl.OracleError if impl is oracle else impl.Vx355Error
            if fits:
                run_agg(impl, [batch], [0], key_types, AGGS, max_rows=1 << 20)
            else:
                with pytest.raises(err, match="integer overflow"):
                    run_agg(impl, [batch], [0], key_types, AGGS, max_rows=1 << 20)


def test_partial_final_chain_and_no_prefix_overflow_equals_the_reference(oracle, vx, monkeypatch):
    """Partial sums near the int64 limits of both signs merge exactly in the FINAL step; where no
    prefix overflows the GPU equals the REFERENCE rule bit for bit (the common case)."""
    monkeypatch.setenv("VX355_AGG_COALESCE_ROWS", "0")
    rng = np.random.default_rng(19)
    n, groups = 300_000, 2_000
    k = rng.integers(0, groups, n).astype(np.int64)
    v = rng.integers(-(1 << 40), 1 << 40, n).astype(np.int64)
    shards = [batch_of([k[i::3], v[i::3]]) for i in range(3)]
    single, _ = run_agg(oracle, [batch_of([k, v])], [0], [abi.BIGINT], AGGS, max_rows=1 << 20)   # reference rule
    parts = []
    for b in shards:
        got, _ = run_agg(vx, [b], [0], [abi.BIGINT], AGGS, abi.STEP_PARTIAL, max_rows=1 << 20)
        parts.append(batch_of([np.asarray(c[0]) for c in got], [np.asarray(c[1]) for c in got]))
    fin = [(abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_COUNT, 2, abi.BIGINT)]
    got, _ = run_agg(vx, parts, [0], [abi.BIGINT], fin, abi.STEP_FINAL, max_rows=1 << 20)
    order = np.argsort(np.asarray(got[0][0]))
    eorder = np.argsort(np.asarray(single[0][0]))
    for c in range(3):
        assert (np.asarray(got[c][0])[order] == np.asarray(single[c][0])[eorder]).all()
    # partial sums +2^62, +2^62, -2^62 of one group: the FINAL total 2^62 fits, whatever the order
    pk = np.array([5, 5, 5, 9], dtype=np.int64)
    ps = np.array([BIG, BIG, -BIG, 1], dtype=np.int64)
    got, _ = run_agg(vx, [batch_of([pk, ps])], [0], [abi.BIGINT], [(abi.AGG_SUM, 1, abi.BIGINT)], abi.STEP_FINAL)
    assert dict(zip(np.asarray(got[0][0]).tolist(), np.asarray(got[1][0]).tolist())) == {5: BIG, 9: 1}
