#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04o
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
b c4s_auto --workload c4 --c4-sparse
VX355_AGG_FOLD_WGS=4 b c4s_auto_wgs4 --workload c4 --c4-sparse
VX355_AGG_DENSE_CHUNK=1 b c4s_chunk1 --workload c4 --c4-sparse
VX355_AGG_HASH_SLOTS=2048 b c4s_2048 --workload c4 --c4-sparse
b c4s_unordered --workload c4 --c4-sparse --c4-unordered
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "c4 or config_4 or billion or sparse" 2>&1 | tail -3
for f in $O/*.err; do echo $f; tail -n 3 $f; done 2>&1 | grep -v amdgpu.ids | head -30
