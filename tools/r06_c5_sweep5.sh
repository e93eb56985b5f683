VX355_JOIN_LDS_BUILD=1 python -m pytest tests/test_gpu_join.py tests/test_gpu_q3_pipeline.py -x -q -m gpu 2>&1 | tail -3
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
run twice VX355_JOIN_WIDE_TWICE=1
run unit8k VX355_JOIN_GROUP_UNIT=8192
run unit8k_twice VX355_JOIN_GROUP_UNIT=8192 VX355_JOIN_WIDE_TWICE=1
run unit8k_twice_wg2 VX355_JOIN_GROUP_UNIT=8192 VX355_JOIN_WIDE_TWICE=1 VX355_JOIN_GROUP_WGS=2
run unit8k_twice_wg3 VX355_JOIN_GROUP_UNIT=8192 VX355_JOIN_WIDE_TWICE=1 VX355_JOIN_GROUP_WGS=3
run narrow_all VX355_JOIN_WIDE_BUILD=0 VX355_JOIN_WIDE=0
run twice_nopf VX355_JOIN_WIDE_TWICE=1 VX355_JOIN_GROUP_PREFETCH=0
run twice_1m VX355_JOIN_WIDE_TWICE=1 VX355_JOIN_SLICE_BYTES=1048576
run twice_4m VX355_JOIN_WIDE_TWICE=1 VX355_JOIN_SLICE_BYTES=4194304
run lds128 VX355_JOIN_LDS_GROUP_BYTES=131072
run lds32 VX355_JOIN_LDS_GROUP_BYTES=32768
