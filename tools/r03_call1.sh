#!/bin/bash
# round 3, GPU call 1: new multi-GPU ABI tests, probe A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dist_abi.py -x -q > gpurun_out/c1_dist_tests.log 2>&1
echo "dist tests rc=$?" >> gpurun_out/c1_dist_tests.log
tail -30 gpurun_out/c1_dist_tests.log
timeout 600 python -m pytest tests/test_gpu_join.py -x -q > gpurun_out/c1_join_tests.log 2>&1
echo "join tests rc=$?" >> gpurun_out/c1_join_tests.log
tail -5 gpurun_out/c1_join_tests.log
for pl in 0 1; do
  VX355_JOIN_PAIR_LOADS=$pl timeout 300 python bench.py --workload q3 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > gpurun_out/c1_q3_pair$pl.json 2> gpurun_out/c1_q3_pair$pl.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/c1_q3_pair$pl.json"))
    print("pair_loads=$pl ms/step", round(d["ms_per_step"], 3), {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.02}, "frac", d["roofline"]["frac"])
except Exception as e:
    print("q3 pair$pl failed", e)
PY
done
