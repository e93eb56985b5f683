#!/bin/bash
# Config 4 scatters: two workgroups per CU on half-size sub-tiles (VX355_AGG_RADIX_HALF_BINS) vs one.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06c4
run() {  # tag, env..., -- flags
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --workload c4 --c4-unordered "$@" --steps 5 --warmup 2 --no-traffic --no-cpu-baseline --no-secondary --detail gpurun_out/r06c4/$tag.json > /dev/null 2> gpurun_out/r06c4/$tag.err
  python - gpurun_out/r06c4/$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d.get("kernels_ms_per_step", {})
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], {n: v for n, v in k.items() if v > 0.3})
PY
}
for h in 0 256 512; do
  run dense_half$h VX355_AGG_RADIX_HALF_BINS=$h --
  run sparse_half$h VX355_AGG_RADIX_HALF_BINS=$h -- --c4-sparse
done
