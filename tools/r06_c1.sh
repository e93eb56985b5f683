#!/bin/bash
# config 1 after the round-6 changes to its step (k_lds_reduce with 16 copies per lane in flight, 128 key-statistics
# blocks, counters read and reset by one launch, the output page written by k_extract into pinned memory).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06c1
mkdir -p $O
for rep in 1 2; do
  python bench.py --workload c1 --steps 50 --warmup 5 --no-traffic --no-cpu-baseline --detail $O/c1_$rep.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
  python - $O/c1_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("c1 ms/step %.4f" % d["ms_per_step"], "check", (d.get("result_check") or {}).get("ok"), {k: round(v, 4) for k, v in d["kernels_ms_per_step"].items() if v > 0.001})
PY
done
python bench.py --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --no-secondary --detail $O/q1.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
python - $O/q1.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("q1 ms/step %.4f" % d["ms_per_step"], "check", (d.get("result_check") or {}).get("ok"))
PY
python -m pytest tests/test_gpu_agg.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py tests/test_gpu_double_sums.py tests/test_gpu_bigint_sums.py tests/test_shim.py tests/test_gpu_async.py tests/test_gpu_threads.py tests/test_gpu_memory_limit.py -q -m gpu -x 2>&1 | tail -3
