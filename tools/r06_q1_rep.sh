#!/bin/bash
# Q1: k_agg_fast with the replica count as a compile-time constant (VX355_AGG_FAST_CONST_REP) vs in a register
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06q1
for rep in 1 2 3; do
  for v in 0 1; do
    VX355_AGG_FAST_CONST_REP=$v python bench.py --steps 30 --warmup 5 --no-traffic --no-cpu-baseline --no-secondary --detail gpurun_out/r06q1/q1_const$v_$rep.json > /dev/null 2>&1
    python - gpurun_out/r06q1/q1_const$v_$rep.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("const_rep", sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % d["roofline"]["kernel_ms_per_step"], "frac %.3f" % d["roofline"]["frac"])
PY
  done
done
for v in 0 1; do
  VX355_AGG_FAST_CONST_REP=$v python bench.py --workload q1x4 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --no-secondary --detail gpurun_out/r06q1/q1x4_const$v.json > /dev/null 2>&1
  python - gpurun_out/r06q1/q1x4_const$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("q1x4 const_rep", sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % d["roofline"]["kernel_ms_per_step"], "frac %.3f" % d["roofline"]["frac"])
PY
done
python -m pytest tests/test_gpu_agg.py tests/test_gpu_full_size.py -q -m gpu -x -k "q1 or Q1 or fast or double_sums" 2>&1 | tail -3
