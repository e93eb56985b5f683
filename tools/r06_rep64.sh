#!/bin/bash
# A replica per lane (REP 64) in up to 128 KB of LDS for plans with many accumulator words (VX355_AGG_LDS_FULL_REPLICAS=0: the
# layouts of before): Q1, four-key Q1, config 1, and the tests that cover the LDS aggregation kernels.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06rep64; mkdir -p $O
for rep in 1 2; do
  for v in 0 1; do
    for wl in q1 q1x4 c1; do
      VX355_AGG_LDS_FULL_REPLICAS=$v python bench.py --workload $wl --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --no-secondary --detail $O/x.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
      python - $O/x.json "full_replicas=$v $wl" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[2], "ms/step %.4f" % d["ms_per_step"], "kernel %.4f" % r["kernel_ms_per_step"], "frac %.3f" % r["frac"])
PY
    done
  done
done
python -m pytest tests/test_gpu_agg.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py tests/test_gpu_double_sums.py tests/test_gpu_bigint_sums.py tests/test_shim.py tests/test_gpu_async.py -q -m gpu -x 2>&1 | tail -2
