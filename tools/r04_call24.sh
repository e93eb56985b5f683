#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04zc
mkdir -p $O
b() { name=$1; shift; timeout 600 python bench.py "$@" --steps 2 --warmup 1 --no-traffic --no-cpu-baseline --no-secondary 2> $O/$name.err | grep '^{"metric"' > $O/$name.json
python - <<PY
import json
try:
    d = json.load(open("$O/$name.json"))
    print("$name", round(d["ms_per_step"], 3), {k: v for k, v in d["kernels_ms_per_step"].items() if 'aggregate' in k})
except Exception as e:
    print("$name FAILED", e)
PY
}
for skip in 0 1 2 3; do
VX355_DEBUG_FOLD_SKIP=$skip b c4_skip$skip --workload c4
done
