#!/bin/bash
# round 4, GPU call 5
mkdir -p gpurun_out/r04e
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_async.py tests/test_gpu_agg.py -x -q -m gpu > gpurun_out/r04e/tests.log 2>&1
tail -4 gpurun_out/r04e/tests.log
for steps in "columns all_gather" "columns counts" "counts columns all_gather" "all_gather columns"; do
  echo "== rccl self: $steps"
  timeout 300 python tests/rccl_self_worker.py $steps > gpurun_out/r04e/rccl_self_$(echo $steps | tr ' ' '_').log 2>&1
  echo "rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r04e/rccl_self_$(echo $steps | tr ' ' '_').log | tail -4
done
timeout 300 python tools/host_timeline.py c1 50 > gpurun_out/r04e/timeline_c1.log 2>&1
tail -3 gpurun_out/r04e/timeline_c1.log
timeout 300 python bench.py --workload c1 --c1-stream --steps 10 --warmup 3 --no-traffic --no-cpu-baseline > gpurun_out/r04e/bench_c1_stream.json 2> gpurun_out/r04e/bench_c1_stream.err
python -c "
import json; d=json.loads(open('gpurun_out/r04e/bench_c1_stream.json').read().strip().splitlines()[-1]); print('c1 stream', d['ms_per_step'], d['workload_info'], d['kernels_ms_per_step'])"
timeout 600 python bench.py --workload q1 --rows 60000000 --host-stream --steps 3 --warmup 1 --no-traffic --no-cpu-baseline --no-secondary > gpurun_out/r04e/bench_q1_stream.json 2> gpurun_out/r04e/bench_q1_stream.err
python -c "
import json; d=json.loads(open('gpurun_out/r04e/bench_q1_stream.json').read().strip().splitlines()[-1]); print('q1 stream', d['ms_per_step'], d['value'], d['workload_info'], d['kernels_ms_per_step'])"
tail -3 gpurun_out/r04e/bench_q1_stream.err
( time timeout 1500 python bench.py > gpurun_out/r04e/bench_default.json 2> gpurun_out/r04e/bench_default.err ) 2> gpurun_out/r04e/bench_default.time
tail -3 gpurun_out/r04e/bench_default.time
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04e/bench_default.json").read().strip().splitlines()[-1])
    print("headline", round(d["ms_per_step"], 3), "ms", d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["measured_ceiling"])
    for k, v in d.get("secondary", {}).items():
        if "error" in v:
            print("  ", k, "ERROR", v["error"]); continue
        r = v.get("roofline") or {}
        print("  ", k, round(v["ms_per_step"], 3), "ms", r.get("kernel"), r.get("frac") and round(r["frac"], 3), "traffic", r.get("traffic"), "cpu", v.get("cpu_baseline", {}).get("value"), v.get("host_ingest"))
except Exception as e:
    print("FAILED", e)
PY
tail -5 gpurun_out/r04e/bench_default.err
