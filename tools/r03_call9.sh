#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
VX355_LOG_SHAPES=1 timeout 600 python -m pytest tests/test_gpu_agg.py tests/test_cpp_consumer.py -m gpu -q -k "nullable or cpp" > gpurun_out/c9_agg_tests.log 2>&1; echo "rc=$?" >> gpurun_out/c9_agg_tests.log; tail -12 gpurun_out/c9_agg_tests.log; grep -h "vx355:" gpurun_out/c9_agg_tests.log | sort | uniq -c | head
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
VX355_LOG_SHAPES=1 VX355_JIT=sync VX355_Q1_NULLS=0.01 timeout 400 python bench.py --workload q1 --steps 5 --warmup 3 --no-cpu-baseline --no-traffic --no-secondary > gpurun_out/c9_q1_nulls.json 2> gpurun_out/c9_q1_nulls.err; summ gpurun_out/c9_q1_nulls.json; grep "vx355:" gpurun_out/c9_q1_nulls.err | sort | uniq -c | head -5
VX355_C5_CHUNKS=1 timeout 300 python bench.py --workload c5 --exchange torch --rows 200000000 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/c9_c5_torch.json 2> gpurun_out/c9_c5_torch.err; summ gpurun_out/c9_c5_torch.json; tail -2 gpurun_out/c9_c5_torch.err
VX355_C5_CHUNKS=1 timeout 300 python bench.py --workload c5 --rows 200000000 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/c9_c5_lib.json 2> gpurun_out/c9_c5_lib.err; summ gpurun_out/c9_c5_lib.json
timeout 300 tools/gather_bench.bin taggroup > gpurun_out/c9_taggroup_bench.txt 2>&1; cat gpurun_out/c9_taggroup_bench.txt
