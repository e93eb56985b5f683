#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_agg.py tests/test_gpu_join.py tests/test_gpu_dist_abi.py tests/test_gpu_q3_pipeline.py -m gpu -q > gpurun_out/c12_tests.log 2>&1; echo "rc=$?" >> gpurun_out/c12_tests.log; tail -8 gpurun_out/c12_tests.log
summ() {
python - "$1" <<'PY'
import json, sys
try:
    lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
    d = json.loads(lines[-1])
    print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.05}, "frac", d["roofline"].get("frac"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
timeout 600 python bench.py --workload c4 --c4-sparse --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/c12_c4_sparse.json 2> gpurun_out/c12_c4_sparse.err; summ gpurun_out/c12_c4_sparse.json
timeout 300 python bench.py --workload q3 --q3-random-probe --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > gpurun_out/c12_q3_random.json 2> gpurun_out/c12_q3_random.err; summ gpurun_out/c12_q3_random.json
VX355_JOIN_PARTITION_FAST=0 timeout 300 python bench.py --workload q3 --q3-random-probe --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > gpurun_out/c12_q3_random_counted.json 2> gpurun_out/c12_q3_random_counted.err; summ gpurun_out/c12_q3_random_counted.json
for ch in 1 4; do
VX355_C5_CHUNKS=$ch timeout 300 python bench.py --workload c5 --rows 200000000 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/c12_c5_ch$ch.json 2> gpurun_out/c12_c5_ch$ch.err; summ gpurun_out/c12_c5_ch$ch.json
done
