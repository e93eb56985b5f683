#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bigint_sums.py -q > gpurun_out/c4_bigint.log 2>&1; echo "rc=$?" >> gpurun_out/c4_bigint.log; tail -25 gpurun_out/c4_bigint.log
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bigint_sums.py > gpurun_out/c4_gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/c4_gpu_tests.log; tail -25 gpurun_out/c4_gpu_tests.log
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
timeout 400 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/c4_c4.json 2> gpurun_out/c4_c4.err; summ gpurun_out/c4_c4.json
timeout 400 python bench.py --workload c4 --c4-unordered --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/c4_c4_unordered.json 2> gpurun_out/c4_c4_unordered.err; summ gpurun_out/c4_c4_unordered.json
