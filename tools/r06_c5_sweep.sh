set -x
python -m pytest tests/test_gpu_join.py -x -q -m gpu -k "regroup or wide or unique" 2>&1 | tail -5
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
run base X=1
run chunks1 VX355_C5_CHUNKS=1
run chunks1_wg4 VX355_C5_CHUNKS=1 VX355_JOIN_GROUP_WGS=4
run chunks1_wg8 VX355_C5_CHUNKS=1 VX355_JOIN_GROUP_WGS=8
run wg2 VX355_JOIN_GROUP_WGS=2
run wg3 VX355_JOIN_GROUP_WGS=3
run wg4 VX355_JOIN_GROUP_WGS=4
run wg8 VX355_JOIN_GROUP_WGS=8
run slice1m VX355_JOIN_SLICE_BYTES=1048576
run slice4m VX355_JOIN_SLICE_BYTES=4194304
run twice VX355_JOIN_WIDE_TWICE=1
run noregroup VX355_JOIN_REGROUP=0
run noregroup_narrow VX355_JOIN_REGROUP=0 VX355_JOIN_WIDE_BUILD=0
