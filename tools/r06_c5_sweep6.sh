python -m pytest tests/test_gpu_join.py tests/test_gpu_dist_abi.py -x -q -m gpu -k "regroup or repartition or wide or dynamic" 2>&1 | tail -3
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
run default X=1
run default2 X=1
run tight VX355_JOIN_WIDE_TIGHT=1
run wg5 VX355_JOIN_GROUP_WGS=5
run wg8 VX355_JOIN_GROUP_WGS=8
run chunks4 VX355_C5_CHUNKS=4
