#!/bin/bash
# round 3, GPU call 2: whole GPU suite on the reworked radix path, c4 benches, probe window A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/c2_gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/c2_gpu_tests.log
tail -15 gpurun_out/c2_gpu_tests.log
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
for w in 0 1; do
  VX355_JOIN_WINDOW=$w timeout 300 python bench.py --workload q3 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > gpurun_out/c2_q3_window$w.json 2> gpurun_out/c2_q3_window$w.err
  summ gpurun_out/c2_q3_window$w.json
done
timeout 400 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/c2_c4.json 2> gpurun_out/c2_c4.err; summ gpurun_out/c2_c4.json
timeout 400 python bench.py --workload c4 --c4-unordered --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/c2_c4_unordered.json 2> gpurun_out/c2_c4_unordered.err; summ gpurun_out/c2_c4_unordered.json
tail -3 gpurun_out/c2_c4.err
