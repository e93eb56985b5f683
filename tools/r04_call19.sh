#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s
mkdir -p $O
b() { name=$1; shift; timeout 600 python bench.py "$@" --steps 2 --warmup 1 --no-traffic --no-cpu-baseline --no-secondary 2> $O/$name.err | grep '^{"metric"' > $O/$name.json
python - <<PY
import json
try:
    d = json.load(open("$O/$name.json"))
    print("$name", round(d["ms_per_step"], 3), {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.5})
except Exception as e:
    print("$name FAILED", e)
PY
}
VX355_AGG_FOLD_WGS=3 b c4s_w6_wgs3 --workload c4 --c4-sparse
VX355_AGG_FOLD_WGS=2 b c4s_w6_wgs2 --workload c4 --c4-sparse
VX355_AGG_FOLD_WGS=3 VX355_DEBUG_FOLD_SKIP=7 b c4s_w6_wgs3_base --workload c4 --c4-sparse
