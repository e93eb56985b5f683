#!/bin/bash
# A/B of the join workloads: product library against velox_amd/variants/libvx355_old.so (k_join_insert claims; later: the probe's cross-tile key prefetch).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06ins; mkdir -p $O
for rep in 1 2; do
  for v in old new; do
    L=$GRAFT_REPO_ROOT/velox_amd/libvx355.so
    [ $v = old ] && L=$GRAFT_REPO_ROOT/velox_amd/variants/libvx355_old.so
    for wl in "q3" "q3full"; do
      VX355_LIB_PATH=$L python bench.py --workload $wl --steps 10 --warmup 3 --no-traffic --no-cpu-baseline --detail $O/x.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
      python - $O/x.json "$v $wl" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], {k: round(v, 3) for k, v in sorted(d["kernels_ms_per_step"].items(), key=lambda x: -x[1])[:4]})
PY
    done
  done
done
python -m pytest tests/test_gpu_join.py tests/test_shim.py -q -m gpu -x 2>&1 | tail -2
