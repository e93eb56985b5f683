#!/bin/bash
# config 4: the radix folds with the next round's records loaded ahead (-DVX355_FOLD_LOOKAHEAD) against the product library.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06fold; mkdir -p $O
for v in main la main la; do
  L=$GRAFT_REPO_ROOT/velox_amd/libvx355.so
  [ $v = la ] && L=$GRAFT_REPO_ROOT/velox_amd/variants/libvx355_la.so
  for wl in "--c4-unordered" "--c4-sparse --c4-unordered"; do
    VX355_LIB_PATH=$L python bench.py --workload c4 $wl --steps 3 --warmup 1 --no-traffic --no-cpu-baseline --detail $O/x.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
    python - $O/x.json "$v $wl" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "check", (d.get("result_check") or {}).get("ok"), {k: round(v, 3) for k, v in sorted(d["kernels_ms_per_step"].items(), key=lambda x: -x[1])[:4]})
PY
  done
done
