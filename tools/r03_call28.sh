#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
summ() {
python - "$1" <<'PY'
import json, sys
try:
    lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
    d = json.loads(lines[-1])
    print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), {k: v for k, v in sorted(d["kernels_ms_per_step"].items(), key=lambda x: -x[1]) if v > 0.3})
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for lib in libvx355.so libvx355_nt.so libvx355.so libvx355_nt.so; do
  VX355_LIB_PATH=$PWD/velox_amd/$lib timeout 300 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/c28_c4_$lib.json 2> gpurun_out/c28_c4_$lib.err; summ gpurun_out/c28_c4_$lib.json
done
for lib in libvx355.so libvx355_nt.so; do
  VX355_LIB_PATH=$PWD/velox_amd/$lib timeout 300 python bench.py --workload c4 --c4-sparse --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/c28_c4s_$lib.json 2> gpurun_out/c28_c4s_$lib.err; summ gpurun_out/c28_c4s_$lib.json
done
