#!/bin/bash
# Round-6 evidence run on the GPU box: the driver's bench line (compact last line + detail file), the other
# bench lines, the rocprofv3 passes (kernel trace with --stats; FETCH_SIZE and WRITE_SIZE in separate
# --kernel-trace-only runs, as MI355X_MICROARCH.md prescribes), smoke(). VX355_SKIP_TESTS=1 skips pytest.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_final
mkdir -p $O
[ -n "$VX355_SKIP_TESTS" ] || ( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/r06_gpu_tests.log 2>&1; tail -3 $O/r06_gpu_tests.log
export TMPDIR=/tmp
run() { name=$1; shift; timeout 900 python bench.py "$@" --detail $O/$name.json 2> $O/$name.err | tail -1 > $O/$name.line.json; python - <<PY
import json
try:
    line = json.load(open("$O/$name.line.json"))
    d = json.load(open("$O/$name.json"))
    print("$name", round(d["ms_per_step"], 3), "ms", "%.3g" % d["value"], "rows/s; line", len(json.dumps(line)), "bytes;",
          {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.02})
    for k, v in d.get("secondary", {}).items():
        r = v.get("roofline") or {}
        print("   ", k, "ERROR " + v["error"] if "error" in v else (round(v["ms_per_step"], 3), "ms", r.get("kernel"), r.get("frac") and round(r["frac"], 3), v.get("host_ingest", {}).get("GBps")))
except Exception as e:
    print("$name FAILED", e)
PY
}
( time python bench.py --detail $O/r06_bench_default.json ) > $O/r06_bench_default.stdout 2> $O/r06_bench_default.err
tail -c 8000 $O/r06_bench_default.stdout | tail -1 > $O/r06_bench_default.line.json
python -c "
import json; l=json.load(open('$O/r06_bench_default.line.json')); print('default line', len(json.dumps(l)), 'bytes', l['value'], l['ms_per_step'], l['roofline']['frac'], l['cpu_baseline']['value'])"
grep real $O/r06_bench_default.err
run r06_bench_c1 --workload c1 --steps 50 --warmup 5
VX355_C1_ROTATE=1 run r06_bench_c1_replayed_input_in_infinity_cache --workload c1 --steps 50 --warmup 5 --no-traffic --no-cpu-baseline
run r06_bench_c1_streamed_host_vectors --workload c1 --c1-stream --steps 10 --warmup 3 --no-traffic
run r06_bench_q1_streamed_host_vectors --workload q1 --rows 60000000 --host-stream --steps 3 --warmup 1 --no-traffic --no-secondary
run r06_bench_q1x4 --workload q1x4 --steps 20 --warmup 5
run r06_bench_q3_join --workload q3
run r06_bench_q3_join_random_probe_order --workload q3 --q3-random-probe
run r06_bench_q3_full_query --workload q3full
run r06_bench_q3_full_query_unfused_filters --workload q3full --unfused --no-traffic --no-cpu-baseline
run r06_bench_c4 --workload c4 --c4-unordered --steps 3 --warmup 1
run r06_bench_c4_first_seen_order --workload c4 --steps 3 --warmup 1 --no-cpu-baseline
run r06_bench_c4_sparse_keys --workload c4 --c4-sparse --c4-unordered --steps 3 --warmup 1
run r06_bench_c4_sparse_keys_first_seen_order --workload c4 --c4-sparse --steps 3 --warmup 1 --no-cpu-baseline
run r06_bench_c5_one_gpu --workload c5 --rows 200000000 --steps 10 --warmup 3
VX355_BENCH_SHARE_GPU=1 run r06_bench_q1_2ranks_sharing_one_gpu --gpus 2 --rows 100000000 --steps 3 --warmup 1 --no-traffic --no-cpu-baseline
VX355_BENCH_SHARE_GPU=1 run r06_bench_c5_2ranks_sharing_one_gpu --gpus 2 --workload c5 --rows 20000000 --steps 2 --warmup 1 --no-traffic --no-cpu-baseline
# rocprofv3
R=$GRAFT_REPO_ROOT
cd /tmp
for wl in q1 q1x4 q3 q3r q3full c1 c4 c4s c4f c4sf c5; do
  args="--steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-traffic --detail ''"
  [ $wl = q3 ] && args="--workload q3 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --detail ''"
  [ $wl = q3r ] && args="--workload q3 --q3-random-probe --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --detail ''"
  [ $wl = q3full ] && args="--workload q3full --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --detail ''"
  [ $wl = c1 ] && args="--workload c1 --steps 50 --warmup 5 --no-cpu-baseline --no-traffic --detail ''"
  [ $wl = c4 ] && args="--workload c4 --c4-unordered --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --detail ''"
  [ $wl = q1x4 ] && args="--workload q1x4 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-traffic --detail ''"
  [ $wl = c4f ] && args="--workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --detail ''"
  [ $wl = c4sf ] && args="--workload c4 --c4-sparse --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --detail ''"
  [ $wl = c5 ] && args="--workload c5 --rows 200000000 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-traffic --detail ''"
  [ $wl = c4s ] && args="--workload c4 --c4-sparse --c4-unordered --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --detail ''"
  eval timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$wl/trace -- python $R/bench.py $args > $R/$O/prof_${wl}_trace.log 2>&1
  eval timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_$wl/fetch -- python $R/bench.py $args > $R/$O/prof_${wl}_fetch.log 2>&1
  eval timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_$wl/write -- python $R/bench.py $args > $R/$O/prof_${wl}_write.log 2>&1
  cd $R
  for d in trace fetch write; do echo "## pass: $d"; python tools/rocprof_summary.py $O/prof_$wl/$d; done > $O/r06_${wl}_rocprofv3_summary.md 2>&1
  find $O/prof_$wl -name "*.csv" -size +5M -delete
  cd /tmp
done
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
ls $O/*.md
