#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04u
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_agg.py -x -q -m gpu -k "first_seen_order_of_many or hashed_folds or dense_folds" 2>&1 | tail -5
b() { name=$1; shift; timeout 600 python bench.py "$@" --steps 3 --warmup 1 --no-traffic --no-cpu-baseline --no-secondary 2> $O/$name.err | grep '^{"metric"' > $O/$name.json
python - <<PY
import json
try:
    d = json.load(open("$O/$name.json"))
    print("$name", round(d["ms_per_step"], 3), {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.02})
except Exception as e:
    print("$name FAILED", e)
PY
}
b c4 --workload c4
VX355_AGG_OWN_SORT_MIN=-1 b c4_rocprim --workload c4
b c4s --workload c4 --c4-sparse
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "c4 or config_4 or billion or sparse" 2>&1 | tail -3
for f in $O/*.err; do echo $f; grep -v amdgpu.ids $f | tail -n 3; done
