#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
summ() {
python - "$1" <<'PY'
import json, sys
try:
    lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
    d = json.loads(lines[-1])
    print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), {k: v for k, v in sorted(d["kernels_ms_per_step"].items(), key=lambda x: -x[1]) if v > 0.05}, d.get("result_check"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
timeout 900 python -m pytest tests/test_gpu_agg.py -x -q -m gpu -k "dense or radix or open_addressing or flush" 2>&1 | tail -3
B="python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic"
VX355_LOG_ALLOC=1 timeout 300 $B --c4-sparse > gpurun_out/c23_c4_sparse.json 2> gpurun_out/c23_c4_sparse.err; summ gpurun_out/c23_c4_sparse.json; grep -c "alloc: fresh" gpurun_out/c23_c4_sparse.err; grep "alloc: fresh" gpurun_out/c23_c4_sparse.err | awk '$5 > 1000000000' | tail -12
timeout 300 $B --c4-sparse --c4-unordered > gpurun_out/c23_c4_sparse_un.json 2> gpurun_out/c23_c4_sparse_un.err; summ gpurun_out/c23_c4_sparse_un.json
VX355_AGG_RADIX_DENSE=0 timeout 300 $B --c4-sparse > gpurun_out/c23_c4_sparse_nodense.json 2> /dev/null; summ gpurun_out/c23_c4_sparse_nodense.json
timeout 300 $B > gpurun_out/c23_c4.json 2> /dev/null; summ gpurun_out/c23_c4.json
