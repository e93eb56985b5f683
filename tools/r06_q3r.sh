#!/bin/bash
# Q3 join in random probe order: k_pp_scatter_fast with the next sub-tile's keys loaded ahead, against the previous library.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06q3r; mkdir -p $O
for rep in 1 2 3; do
  for v in old new; do
    L=$GRAFT_REPO_ROOT/velox_amd/libvx355.so
    [ $v = old ] && L=$GRAFT_REPO_ROOT/velox_amd/variants/libvx355_old.so
    VX355_LIB_PATH=$L python bench.py --workload q3 --q3-random-probe --steps 10 --warmup 3 --no-traffic --no-cpu-baseline --detail $O/q3r_${v}_$rep.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
    python - $O/q3r_${v}_$rep.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("q3r", sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "check", (d.get("result_check") or {}).get("ok"), {k: round(v, 3) for k, v in sorted(d["kernels_ms_per_step"].items(), key=lambda x: -x[1])[:4]})
PY
  done
done
python -m pytest tests/test_gpu_join.py -q -m gpu -x -k "partitioned or input_filter" 2>&1 | tail -2
