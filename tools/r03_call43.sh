#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
summ() {
python - "$1" <<'PY'
import json, sys
try:
    lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
    d = json.loads(lines[-1])
    print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), {k: v for k, v in sorted(d["kernels_ms_per_step"].items(), key=lambda x: -x[1]) if v > 0.05}, "frac", round(d["roofline"]["frac"], 3))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for lib in libvx355.so libvx355_u16.so libvx355.so libvx355_u16.so; do
  VX355_LIB_PATH=$PWD/velox_amd/$lib timeout 300 python bench.py --workload q3 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > gpurun_out/c43_q3_$lib.json 2> gpurun_out/c43_q3_$lib.err; summ gpurun_out/c43_q3_$lib.json
done
