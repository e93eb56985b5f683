"""Host-side time of each operator call of one step (create / add_input / no_more_input / get_output /
destroy) for an aggregation workload of bench.py, next to the kernels' times: where does a step go
that is much longer than its kernels?  python tools/host_timeline.py c1|q1|q1x4 [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from velox_amd import ops, abi
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "c1"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
ops.init(0)
cls, rows = bench.WORKLOADS[name]
wl = cls(torch, rows, dev, seed=1234)


def make():
    if name == "c1":
        return ops.HashAggregation([0], [abi.BIGINT], wl.AGGS, abi.STEP_SINGLE), wl.batch
    if name == "q1":
        o = ops.HashAggregation(bench.Q1_KEYS[0], bench.Q1_KEYS[1], wl.FUSED_AGGS, abi.STEP_SINGLE)
        o.set_fused_input(bench.Q1_TERMS, bench.Q1_PROJ)
        return o, wl.scan
    return wl.operator(abi.STEP_SINGLE), wl.scan


for _ in range(5):
    wl.step()
ops.synchronize()
for profiled in (False, True):
    acc = {}

    def t(label, fn):
        t0 = time.perf_counter()
        r = fn()
        acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        return r
    ops.profile_reset()
    ops.profile_enable(profiled)
    for _ in range(N):
        op, batch = t("create", make)
        t("add_input", lambda: op.add_input(batch))
        t("no_more_input", op.no_more_input)
        t("get_output", lambda: ops.collect_output(op, 4096))
        holder = [op]
        del op
        t("destroy", holder.clear)
    ops.profile_enable(False)
    print("profiling", profiled, {k: round(v / N * 1e3, 4) for k, v in acc.items()}, "ms per step; total",
          round(sum(acc.values()) / N * 1e3, 4))
    if profiled:
        print({k: (round(v[0] / N, 4), round(v[1] / N, 2)) for k, v in sorted(ops.profile().items())})
