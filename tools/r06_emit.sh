#!/bin/bash
# k_emit with four iterations' loads ahead, against the previous library (config 5 and the Q3 join).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06emit; mkdir -p $O
for rep in 1 2; do
  for v in old new; do
    L=$GRAFT_REPO_ROOT/velox_amd/libvx355.so
    [ $v = old ] && L=$GRAFT_REPO_ROOT/velox_amd/variants/libvx355_old.so
    VX355_LIB_PATH=$L python bench.py --workload c5 --rows 200000000 --steps 10 --warmup 3 --no-traffic --no-cpu-baseline --detail $O/c5_${v}_$rep.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
    python - $O/c5_${v}_$rep.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("c5", sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "check", (d.get("result_check") or {}).get("ok"), {k: round(v, 3) for k, v in sorted(d["kernels_ms_per_step"].items(), key=lambda x: -x[1])[:5]})
PY
  done
done
python -m pytest tests/test_gpu_join.py tests/test_gpu_dist_abi.py tests/test_shim.py -q -m gpu -x 2>&1 | tail -2
