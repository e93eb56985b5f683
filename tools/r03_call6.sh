#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_agg.py tests/test_gpu_double_sums.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py tests/test_gpu_bigint_sums.py tests/test_gpu_kernels.py tests/test_gpu_join.py -m gpu -q > gpurun_out/c6_gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/c6_gpu_tests.log; tail -15 gpurun_out/c6_gpu_tests.log
summ() {
python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.05}, "frac", d["roofline"].get("frac"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
timeout 400 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/c6_c4.json 2> gpurun_out/c6_c4.err; summ gpurun_out/c6_c4.json
VX355_AGG_RADIX_TILE2=262144 timeout 400 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/c6_c4_tile2.json 2> gpurun_out/c6_c4_tile2.err; summ gpurun_out/c6_c4_tile2.json
VX355_AGG_RADIX_OPTIMISTIC=0 timeout 400 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/c6_c4_exact.json 2> gpurun_out/c6_c4_exact.err; summ gpurun_out/c6_c4_exact.json
timeout 400 python bench.py --workload c4 --c4-unordered --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/c6_c4_unordered.json 2> gpurun_out/c6_c4_unordered.err; summ gpurun_out/c6_c4_unordered.json
timeout 300 python bench.py --workload q3 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > gpurun_out/c6_q3.json 2> gpurun_out/c6_q3.err; summ gpurun_out/c6_q3.json
timeout 300 python bench.py --workload q3 --q3-random-probe --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > gpurun_out/c6_q3_random.json 2> gpurun_out/c6_q3_random.err; summ gpurun_out/c6_q3_random.json
timeout 300 python bench.py --workload c5 --rows 200000000 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/c6_c5.json 2> gpurun_out/c6_c5.err; summ gpurun_out/c6_c5.json; tail -3 gpurun_out/c6_c5.err
