python -m pytest tests/test_gpu_join.py tests/test_gpu_dist_abi.py -x -q -m gpu 2>&1 | tail -5
run() { # name env...
  name=$1; shift
  env "$@" python bench.py --workload c5 --rows 200000000 --steps 5 --warmup 2 --no-traffic --no-cpu-baseline --detail gpurun_out/c5_$name.json > /dev/null 2>&1
  python - <<PY
import json
d=json.load(open("gpurun_out/c5_$name.json"))
k=d["kernels_ms_per_step"]
print("$name", round(d["ms_per_step"],2), {x:round(v,2) for x,v in k.items() if v>0.1}, d["result_check"]["ok"])
PY
}
run chunks1 X=1
run chunks1_nopf VX355_JOIN_GROUP_PREFETCH=0
run chunks1_twice VX355_JOIN_WIDE_TWICE=1
run chunks4 VX355_C5_CHUNKS=4
run chunks4_nopf VX355_C5_CHUNKS=4 VX355_JOIN_GROUP_PREFETCH=0
run chunks4_twice VX355_C5_CHUNKS=4 VX355_JOIN_WIDE_TWICE=1
run chunks1_wg4 VX355_JOIN_GROUP_WGS=4
run chunks1_wg2 VX355_JOIN_GROUP_WGS=2
