#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04p
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_agg.py -x -q -m gpu -k "radix or dense_folds or sorted or optimistic or unordered" 2>&1 | tail -5
b() { name=$1; shift; timeout 600 python bench.py "$@" --steps 3 --warmup 1 --no-traffic --no-cpu-baseline --no-secondary 2> $O/$name.err | grep '^{"metric"' > $O/$name.json
python - <<PY
import json
try:
    d = json.load(open("$O/$name.json"))
    print("$name", round(d["ms_per_step"], 3), {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.02}, d.get("result_check"))
except Exception as e:
    print("$name FAILED", e)
PY
}
b c4s --workload c4 --c4-sparse
b c4 --workload c4
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "c4 or config_4 or billion or sparse" 2>&1 | tail -3
