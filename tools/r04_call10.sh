#!/bin/bash
mkdir -p gpurun_out/r04j
cd "$GRAFT_REPO_ROOT"
for v in "--c4-sparse" ""; do
  timeout 300 python bench.py --workload c4 $v --steps 3 --warmup 1 --no-traffic --no-cpu-baseline > gpurun_out/r04j/bench_c4$v.json 2> gpurun_out/r04j/bench_c4$v.err
  python -c "
import json,sys; d=json.loads(open('gpurun_out/r04j/bench_c4$v.json').read().strip().splitlines()[-1]); print('c4 $v', round(d['ms_per_step'],2), d['kernels_ms_per_step'])"
done
timeout 900 python -m pytest tests/test_gpu_agg.py tests/test_gpu_join.py tests/test_gpu_kernels.py tests/test_gpu_dist_abi.py -x -q -m gpu > gpurun_out/r04j/tests.log 2>&1
tail -6 gpurun_out/r04j/tests.log
