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
run chunks1_twice VX355_C5_CHUNKS=1 VX355_JOIN_WIDE_TWICE=1
run chunks1_4m VX355_C5_CHUNKS=1 VX355_JOIN_SLICE_BYTES=4194304
run chunks1_1m VX355_C5_CHUNKS=1 VX355_JOIN_SLICE_BYTES=1048576
run chunks2 VX355_C5_CHUNKS=2
bash tools/r06_c5_counters.sh chunks1 VX355_C5_CHUNKS=1
bash tools/r06_c5_counters.sh chunks4 VX355_C5_CHUNKS=4
