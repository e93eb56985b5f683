#!/bin/bash
# Round-4 evidence run on the GPU box: the driver's bench line, the other bench lines, the rocprofv3
# passes (kernel trace with --stats; FETCH_SIZE and WRITE_SIZE in separate --kernel-trace-only runs, as
# MI355X_MICROARCH.md prescribes), the RCCL self path under the kernel trace, smoke().
cd "$GRAFT_REPO_ROOT" || exit 1
[ -n "$VX355_SKIP_TESTS" ] || ( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/r04_final_tests.log 2>&1; tail -3 gpurun_out/r04_final_tests.log
O=gpurun_out/r04_final
mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; timeout 900 python bench.py "$@" 2> $O/$name.err | grep '^{"metric"' | head -1 > $O/$name.json; python - <<PY
import json
try:
    d = json.load(open("$O/$name.json"))
    print("$name", round(d["ms_per_step"], 3), "ms", "%.3g" % d["value"], "rows/s", {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.02})
    for k, v in d.get("secondary", {}).items():
        r = v.get("roofline") or {}
        print("   ", k, "ERROR " + v["error"] if "error" in v else (round(v["ms_per_step"], 3), "ms", r.get("kernel"), r.get("frac") and round(r["frac"], 3), v.get("host_ingest", {}).get("GBps")))
except Exception as e:
    print("$name FAILED", e)
PY
}
run r04_bench_default --steps 20 --warmup 5
run r04_bench_c1 --workload c1 --steps 50 --warmup 5
VX355_C1_NULLS=0.5 run r04_bench_c1_half_null_values --workload c1 --steps 50 --warmup 5 --no-traffic
VX355_AGG_SCRATCH_FLUSH=0 run r04_bench_c1_atomics_flush --workload c1 --steps 50 --warmup 5 --no-traffic --no-cpu-baseline
run r04_bench_c1_streamed_host_vectors --workload c1 --c1-stream --steps 10 --warmup 3 --no-traffic
VX355_INGEST_PARALLEL=0 run r04_bench_c1_streamed_host_vectors_serial_ingest --workload c1 --c1-stream --steps 10 --warmup 3 --no-traffic --no-cpu-baseline
run r04_bench_q1_streamed_host_vectors --workload q1 --rows 60000000 --host-stream --steps 3 --warmup 1 --no-traffic --no-secondary
run r04_bench_q1x4 --workload q1x4 --steps 20 --warmup 5
run r04_bench_q1_unfused --unfused --no-secondary --no-traffic
run r04_bench_q3_full_query --workload q3full --no-traffic
run r04_bench_c4 --workload c4 --steps 3 --warmup 1
run r04_bench_c4_unordered_output --workload c4 --c4-unordered --steps 3 --warmup 1 --no-traffic --no-cpu-baseline
run r04_bench_c4_sparse_keys --workload c4 --c4-sparse --steps 3 --warmup 1
run r04_bench_c4_sparse_keys_unordered_output --workload c4 --c4-sparse --c4-unordered --steps 3 --warmup 1 --no-traffic --no-cpu-baseline
VX355_C5_CHUNKS=1 run r04_bench_c5_one_gpu --workload c5 --rows 200000000 --steps 5 --warmup 2 --no-traffic
VX355_COMM_FORCE_RCCL=1 VX355_C5_CHUNKS=4 run r04_bench_c5_one_gpu_through_rccl --workload c5 --rows 200000000 --steps 3 --warmup 1 --no-traffic --no-cpu-baseline
# rocprofv3
R=$GRAFT_REPO_ROOT
cd /tmp
for wl in q1 q1x4 c1 c4 c4s; do
  args="--steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-traffic"
  [ $wl = q1x4 ] && args="--workload q1x4 $args"
  [ $wl = c1 ] && args="--workload c1 --steps 50 --warmup 5 --no-cpu-baseline --no-traffic"
  [ $wl = c4 ] && args="--workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic"
  [ $wl = c4s ] && args="--workload c4 --c4-sparse --steps 2 --warmup 1 --no-cpu-baseline --no-traffic"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$wl/trace -- python $R/bench.py $args > $R/$O/prof_${wl}_trace.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_$wl/fetch -- python $R/bench.py $args > $R/$O/prof_${wl}_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_$wl/write -- python $R/bench.py $args > $R/$O/prof_${wl}_write.log 2>&1
  cd $R
  for d in trace fetch write; do echo "## pass: $d"; python tools/rocprof_summary.py $O/prof_$wl/$d; done > $O/r04_${wl}_rocprofv3_summary.md 2>&1
  find $O/prof_$wl -name "*.csv" -size +5M -delete
  cd /tmp
done
# the RCCL entry points on one GPU (VX355_COMM_FORCE_RCCL=1): which device kernels ran
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_rccl/trace -- python $R/tests/rccl_self_worker.py > $R/$O/prof_rccl_trace.log 2>&1
cd $R
{ echo "## VX355_COMM_FORCE_RCCL=1: tests/rccl_self_worker.py under rocprofv3 --kernel-trace --stats"; tail -3 $O/prof_rccl_trace.log; python tools/rocprof_summary.py $O/prof_rccl/trace; } > $O/r04_rccl_self_rocprofv3_summary.md 2>&1
find $O/prof_rccl -name "*.csv" -size +5M -delete
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
ls $O/*.md
