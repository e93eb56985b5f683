#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/c5_gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/c5_gpu_tests.log; tail -25 gpurun_out/c5_gpu_tests.log
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
for wgs in 2 3 4; do
VX355_AGG_FOLD_WGS=$wgs timeout 400 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/c5_c4_fold$wgs.json 2> gpurun_out/c5_c4_fold$wgs.err; summ gpurun_out/c5_c4_fold$wgs.json
done
timeout 400 python bench.py --workload c4 --c4-unordered --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/c5_c4_unordered.json 2> gpurun_out/c5_c4_unordered.err; summ gpurun_out/c5_c4_unordered.json
for lib in libvx355.so libvx355_u8.so; do
VX355_LIB_PATH=$PWD/velox_amd/$lib timeout 300 python bench.py --workload q3 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > gpurun_out/c5_q3_$lib.json 2> gpurun_out/c5_q3_$lib.err; summ gpurun_out/c5_q3_$lib.json
done
