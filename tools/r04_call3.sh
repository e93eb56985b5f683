#!/bin/bash
# round 4, GPU call 3: C1 fixes, RCCL self path diagnostics, whole default bench line
mkdir -p gpurun_out/r04c
cd "$GRAFT_REPO_ROOT"
export VX355_LOG_SHAPES=1
timeout 900 python -m pytest tests/test_gpu_agg.py tests/test_gpu_kernels.py tests/test_gpu_q3_pipeline.py -x -q -m gpu -s > gpurun_out/r04c/tests_agg.log 2>&1
tail -5 gpurun_out/r04c/tests_agg.log
timeout 600 python tests/rccl_self_worker.py > gpurun_out/r04c/rccl_self.log 2>&1
tail -25 gpurun_out/r04c/rccl_self.log
timeout 300 python tools/host_timeline.py c1 50 > gpurun_out/r04c/timeline_c1.log 2>&1
tail -4 gpurun_out/r04c/timeline_c1.log
timeout 300 python tools/host_timeline.py q1x4 10 > gpurun_out/r04c/timeline_q1x4.log 2>&1
tail -4 gpurun_out/r04c/timeline_q1x4.log
unset VX355_LOG_SHAPES
( time timeout 1500 python bench.py > gpurun_out/r04c/bench_default.json 2> gpurun_out/r04c/bench_default.err ) 2> gpurun_out/r04c/bench_default.time
tail -3 gpurun_out/r04c/bench_default.time
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04c/bench_default.json").read().strip().splitlines()[-1])
    print("headline", round(d["ms_per_step"], 3), "ms", d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["measured_ceiling"])
    for k, v in d.get("secondary", {}).items():
        if "error" in v:
            print("  ", k, "ERROR", v["error"]); continue
        r = v["roofline"]
        print("  ", k, round(v["ms_per_step"], 3), "ms", r["kernel"], r["frac"] and round(r["frac"], 3), "traffic", r["traffic"], r["traffic_source"] and r["traffic_source"][:20], "cpu", v.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print("FAILED", e)
PY
tail -5 gpurun_out/r04c/bench_default.err
