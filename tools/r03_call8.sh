#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_join.py tests/test_gpu_dist_abi.py tests/test_gpu_fuzz.py tests/test_cpp_consumer.py -m gpu -q > gpurun_out/c8_gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/c8_gpu_tests.log; tail -15 gpurun_out/c8_gpu_tests.log
timeout 600 python -m pytest tests/test_gpu_agg.py -m gpu -q -k "nullable or hiprtc or async or dictionary" > gpurun_out/c8_agg_tests.log 2>&1; echo "rc=$?" >> gpurun_out/c8_agg_tests.log; tail -15 gpurun_out/c8_agg_tests.log
summ() {
python - "$1" <<'PY'
import json, sys
try:
    lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
    d = json.loads(lines[-1])
    print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.05}, "frac", d["roofline"].get("frac"), "stdout lines", sum(1 for _ in open(sys.argv[1])))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for ch in 1 4; do for lib in libvx355.so libvx355_u4.so; do
VX355_C5_CHUNKS=$ch VX355_LIB_PATH=$PWD/velox_amd/$lib timeout 300 python bench.py --workload c5 --rows 200000000 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/c8_c5_ch${ch}_$lib.json 2> gpurun_out/c8_c5_ch${ch}_$lib.err; summ gpurun_out/c8_c5_ch${ch}_$lib.json
done; done
VX355_JIT=sync VX355_Q1_NULLS=0.01 timeout 400 python bench.py --workload q1 --steps 5 --warmup 3 --no-cpu-baseline --no-traffic --no-secondary > gpurun_out/c8_q1_nulls.json 2> gpurun_out/c8_q1_nulls.err; summ gpurun_out/c8_q1_nulls.json; tail -2 gpurun_out/c8_q1_nulls.err
