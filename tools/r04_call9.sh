#!/bin/bash
mkdir -p gpurun_out/r04i
cd "$GRAFT_REPO_ROOT"
for v in "" "--c4-sparse" "--c4-unordered"; do
  timeout 300 python bench.py --workload c4 $v --steps 3 --warmup 1 --no-traffic --no-cpu-baseline > gpurun_out/r04i/bench_c4$v.json 2> gpurun_out/r04i/bench_c4$v.err
  echo "rc=$?"; tail -2 gpurun_out/r04i/bench_c4$v.err
  python -c "
import json,sys; d=json.loads(open('gpurun_out/r04i/bench_c4$v.json').read().strip().splitlines()[-1]); print('c4 $v', round(d['ms_per_step'],2), d['kernels_ms_per_step'])"
done
VX355_AGG_RADIX_OPTIMISTIC1=0 timeout 300 python bench.py --workload c4 --steps 3 --warmup 1 --no-traffic --no-cpu-baseline > gpurun_out/r04i/bench_c4_counted.json 2>/dev/null
python -c "
import json,sys; d=json.loads(open('gpurun_out/r04i/bench_c4_counted.json').read().strip().splitlines()[-1]); print('c4 counted level 1', round(d['ms_per_step'],2), d['kernels_ms_per_step'])"
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "c4" > gpurun_out/r04i/tests_full.log 2>&1
tail -4 gpurun_out/r04i/tests_full.log
