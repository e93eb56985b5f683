#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04zf
mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_agg.py tests/test_gpu_fuzz.py tests/test_gpu_double_sums.py tests/test_gpu_bigint_sums.py -x -q -m gpu 2>&1 | tail -3 )
b() { name=$1; shift; timeout 600 python bench.py "$@" --steps 3 --warmup 1 --no-traffic --no-cpu-baseline --no-secondary 2> $O/$name.err | grep '^{"metric"' > $O/$name.json
python - <<PY
import json
try:
    d = json.load(open("$O/$name.json"))
    print("$name", round(d["ms_per_step"], 3), {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.3})
except Exception as e:
    print("$name FAILED", e)
PY
}
b c4s --workload c4 --c4-sparse
b c4 --workload c4
b q1 --workload q1 --steps 10
