#!/bin/bash
# Join workloads after k_build_append_flat (round 6).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06join
mkdir -p $O
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "check", (d.get("result_check") or {}).get("ok"),
      {k: round(v, 3) for k, v in sorted(d["kernels_ms_per_step"].items(), key=lambda x: -x[1])[:6]})
PY
}
python bench.py --workload q3 --no-traffic --no-cpu-baseline --detail $O/q3.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
show $O/q3.json q3
python bench.py --workload q3full --no-traffic --no-cpu-baseline --detail $O/q3full.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
show $O/q3full.json q3full
python bench.py --workload c5 --rows 200000000 --steps 10 --warmup 3 --no-traffic --no-cpu-baseline --detail $O/c5.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
show $O/c5.json c5
python -m pytest tests/test_gpu_join.py tests/test_gpu_dist_abi.py tests/test_shim.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -3
python bench.py --workload c1 --steps 50 --warmup 5 --no-traffic --no-cpu-baseline --detail $O/c1.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
show $O/c1.json c1
python -m pytest tests/test_gpu_agg.py tests/test_gpu_async.py tests/test_gpu_full_size.py -q -m gpu -x 2>&1 | tail -2
