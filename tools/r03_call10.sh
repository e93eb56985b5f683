#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_agg.py -m gpu -q -x -k "open_addressing" > gpurun_out/c10_hashed_tests.log 2>&1; echo "rc=$?" >> gpurun_out/c10_hashed_tests.log; tail -30 gpurun_out/c10_hashed_tests.log
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
timeout 600 python bench.py --workload c4 --c4-sparse --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/c10_c4_sparse.json 2> gpurun_out/c10_c4_sparse.err; summ gpurun_out/c10_c4_sparse.json; tail -3 gpurun_out/c10_c4_sparse.err
timeout 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_agg.py tests/test_gpu_double_sums.py tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/c10_agg_tests.log 2>&1; echo "rc=$?" >> gpurun_out/c10_agg_tests.log; tail -12 gpurun_out/c10_agg_tests.log
