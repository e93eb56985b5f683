# Re-measures every workload bench.py knows on the GPU box; JSON lines land in gpurun_out/refresh/.
# ROUND=r03 sh tools/refresh_profiles.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
P=${ROUND:-r03}
cd $R
O=gpurun_out/refresh
mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py "$@" 2> $O/$name.err | grep '^{"metric"' | head -1 > $O/$name.json; python - <<PY
import json
try:
    d = json.load(open("$O/$name.json"))
    print("$name", round(d["ms_per_step"], 3), "ms", "%.3g" % d["value"], "rows/s", {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.3})
    for k, v in d.get("secondary", {}).items():
        print("   ", k, round(v["ms_per_step"], 3), "ms", v["roofline"]["kernel"], round(v["roofline"]["achieved"] or 0), "GB/s")
except Exception as e:
    print("$name FAILED", e)
PY
}
run ${P}_bench_default --steps 20 --warmup 5
# (hiprtc compiles in the background by default: a bench of 25 short-lived operators would be over before
# the specialised kernel is ready, so this line waits for it)
VX355_JIT=sync run ${P}_bench_q1_unfused --unfused --no-secondary --no-traffic
run ${P}_bench_c1 --workload c1 --no-traffic
run ${P}_bench_c1_streamed_host_batches --workload c1 --c1-stream --no-traffic
run ${P}_bench_q3_full_query --workload q3full --no-traffic
VX355_JOIN_PARTITION=0 run ${P}_bench_q3_join_random_direct_probe --workload q3 --q3-random-probe --no-traffic --no-cpu-baseline
run ${P}_bench_c4 --workload c4 --steps 3 --warmup 1
run ${P}_bench_c4_unordered_output --workload c4 --c4-unordered --steps 3 --warmup 1 --no-traffic --no-cpu-baseline
run ${P}_bench_c4_sparse_keys --workload c4 --c4-sparse --steps 3 --warmup 1 --no-traffic
run ${P}_bench_c4_sparse_keys_unordered_output --workload c4 --c4-sparse --c4-unordered --steps 3 --warmup 1 --no-traffic --no-cpu-baseline
VX355_C5_CHUNKS=1 run ${P}_bench_c5_one_gpu --workload c5 --rows 200000000 --steps 5 --warmup 2 --no-traffic
run ${P}_bench_q3_join --workload q3 --steps 20 --warmup 5
run ${P}_bench_q3_join_random_probe_order --workload q3 --q3-random-probe --steps 10 --warmup 3
VX355_Q1_NULLS=0.01 run ${P}_bench_q1_nullable_discount --workload q1 --steps 10 --warmup 3 --no-traffic --no-secondary --no-cpu-baseline
# two ranks SHARING the one GPU of this box: the launcher, the in-library RCCL exchange and the
# merge run end to end (a functional record, not a scaling number: both ranks use the same HBM)
VX355_BENCH_SHARE_GPU=1 run ${P}_bench_q1_2ranks_sharing_one_gpu --gpus 2 --rows 100000000 --steps 5 --warmup 2
VX355_BENCH_SHARE_GPU=1 run ${P}_bench_c5_2ranks_sharing_one_gpu --gpus 2 --workload c5 --rows 50000000 --steps 3 --warmup 1
VX355_C1_NULLS=0.5 run ${P}_bench_c1_half_null_values --workload c1 --no-traffic
