#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_cpp_consumer.py tests/test_gpu_async.py tests/test_abi_load.py -x -q 2>&1 | tail -4
for mode in 0 1; do
  VX355_C1_ASYNC=$mode timeout 300 python bench.py --workload c1 --c1-stream --no-traffic --no-cpu-baseline > gpurun_out/c45_c1_stream_$mode.json 2> gpurun_out/c45_c1_stream_$mode.err
  python - gpurun_out/c45_c1_stream_$mode.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], round(d["ms_per_step"], 3), d["workload_info"])
PY
done
