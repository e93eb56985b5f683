"""Where do the ~0.7 ms between Q1's kernels and its step time go? Host-side timing of each call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from velox_amd import ops, abi
import bench

dev = torch.device("cuda:0")
ops.init(0)
wl = bench.Q1(torch, 600_037_902, dev, seed=1234)
wl.fused = True
for _ in range(5):
    wl.step()
ops.synchronize()
acc = {}
def t(name, fn):
    t0 = time.perf_counter(); r = fn(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
N = 20
ops.profile_reset(); ops.profile_enable(True)
for _ in range(N):
    def create():
        o = ops.HashAggregation(bench.Q1_KEYS[0], bench.Q1_KEYS[1], wl.FUSED_AGGS, abi.STEP_SINGLE)
        o.set_fused_input(bench.Q1_TERMS, bench.Q1_PROJ)
        return o
    op = t("create", create)
    t("add_input", lambda: op.add_input(wl.scan))
    t("no_more_input", lambda: op.no_more_input())
    t("get_output", lambda: ops.collect_output(op, 1024))
    def destroy():
        nonlocal_op[0] = None
    nonlocal_op = [op]
    del op
    t("destroy", destroy)
ops.profile_enable(False)
print({k: round(v / N * 1e3, 3) for k, v in acc.items()}, "ms per step; total", round(sum(acc.values()) / N * 1e3, 3))
print({k: (round(v[0] / N, 4), v[1] // N) for k, v in ops.profile().items()})
