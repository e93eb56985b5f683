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
run slice2m VX355_JOIN_SLICE_BYTES=2097152
run lds128 VX355_JOIN_LDS_GROUP_BYTES=131072
run default2 X=1
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
