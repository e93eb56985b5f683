#!/bin/bash
# End of round 6: config 4 again after the dense fold's look-ahead loads (bench lines, rocprofv3 passes of the dense forms), and the tests that cover the folds.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_final
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_agg.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -2
run() { name=$1; shift; timeout 900 python bench.py "$@" --detail $O/$name.json 2> $O/$name.err | tail -1 > $O/$name.line.json; python - <<PY
import json
d = json.load(open("$O/$name.json"))
print("$name", round(d["ms_per_step"], 3), "ms", (d.get("result_check") or {}).get("ok"), {k: round(v, 3) for k, v in d["kernels_ms_per_step"].items() if v > 0.5})
PY
}
run r06_bench_c4 --workload c4 --c4-unordered --steps 3 --warmup 1
run r06_bench_c4_first_seen_order --workload c4 --steps 3 --warmup 1 --no-cpu-baseline
run r06_bench_c4_sparse_keys --workload c4 --c4-sparse --c4-unordered --steps 3 --warmup 1
run r06_bench_c4_sparse_keys_first_seen_order --workload c4 --c4-sparse --steps 3 --warmup 1 --no-cpu-baseline
R=$GRAFT_REPO_ROOT
cd /tmp
for wl in c4 c4f; do
  args="--workload c4 --c4-unordered --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --detail ''"
  [ $wl = c4f ] && args="--workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --detail ''"
  rm -rf $R/$O/prof_$wl
  eval timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$wl/trace -- python $R/bench.py $args > $R/$O/prof_${wl}_trace.log 2>&1
  eval timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_$wl/fetch -- python $R/bench.py $args > $R/$O/prof_${wl}_fetch.log 2>&1
  eval timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_$wl/write -- python $R/bench.py $args > $R/$O/prof_${wl}_write.log 2>&1
  cd $R
  for d in trace fetch write; do echo "## pass: $d"; python tools/rocprof_summary.py $O/prof_$wl/$d; done > $O/r06_${wl}_rocprofv3_summary.md 2>&1
  find $O/prof_$wl -name "*.csv" -size +5M -delete
  grep "k_rp_aggregate" $O/r06_${wl}_rocprofv3_summary.md | head -4
  cd /tmp
done
