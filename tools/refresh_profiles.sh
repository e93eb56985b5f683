# Re-measures every workload bench.py knows on the GPU box; JSON lines land in gpurun_out/refresh/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/refresh
mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py "$@" 2> $O/$name.err | grep '^{"metric"' | head -1 > $O/$name.json; python - <<PY
import json
try:
    d = json.load(open("$O/$name.json"))
    print("$name", round(d["ms_per_step"], 3), "ms", "%.3g" % d["value"], "rows/s", {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.3})
except Exception as e:
    print("$name FAILED", e)
PY
}
run r01_bench_q1_sf100
run r01_bench_q1_unfused --unfused
run r01_bench_c1 --workload c1
run r01_bench_c1_streamed_host_batches --workload c1 --c1-stream
run r01_bench_q3_join --workload q3
run r01_bench_q3_join_random_probe_order --workload q3 --q3-random-probe
run r01_bench_q3_full_query --workload q3full
run r01_bench_c4 --workload c4 --steps 3 --warmup 1
run r01_bench_c4_sparse_keys --workload c4 --c4-sparse --steps 3 --warmup 1
VX355_AGG_RADIX_MIN_ROWS=-1 run r01_bench_c4_atomics_only --workload c4 --steps 2 --warmup 1 --no-cpu-baseline
run r01_bench_c5_one_gpu --workload c5 --rows 200000000 --steps 3 --warmup 1
