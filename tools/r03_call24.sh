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
timeout 1200 python -m pytest tests/test_gpu_join.py tests/test_gpu_dist_abi.py -x -q -m gpu 2>&1 | tail -5
B="python bench.py --workload c5 --rows 200000000 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic"
VX355_C5_CHUNKS=1 timeout 300 $B > gpurun_out/c24_c5_ch1.json 2> gpurun_out/c24_c5_ch1.err; summ gpurun_out/c24_c5_ch1.json; tail -2 gpurun_out/c24_c5_ch1.err
timeout 300 $B > gpurun_out/c24_c5_ch4.json 2> gpurun_out/c24_c5_ch4.err; summ gpurun_out/c24_c5_ch4.json
VX355_JOIN_WIDE=0 VX355_C5_CHUNKS=1 timeout 300 $B > gpurun_out/c24_c5_nowide.json 2> /dev/null; summ gpurun_out/c24_c5_nowide.json
