#!/bin/bash
# refresh of the config-4 evidence and the driver's line after the last kernel change
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_final
mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; timeout 900 python bench.py "$@" 2> $O/$name.err | grep '^{"metric"' | head -1 > $O/$name.json; python - <<PY
import json
d = json.load(open("$O/$name.json"))
print("$name", round(d["ms_per_step"], 3), "ms", {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.3})
for k, v in d.get("secondary", {}).items():
    r = v.get("roofline") or {}
    print("   ", k, round(v["ms_per_step"], 3), r.get("kernel"), r.get("frac") and round(r["frac"], 3), r.get("traffic"), (v.get("host_ingest") or {}).get("GBps"))
PY
}
run r04_bench_default --steps 20 --warmup 5
run r04_bench_c4 --workload c4 --steps 3 --warmup 1
run r04_bench_c4_unordered_output --workload c4 --c4-unordered --steps 3 --warmup 1 --no-traffic --no-cpu-baseline
run r04_bench_c4_sparse_keys --workload c4 --c4-sparse --steps 3 --warmup 1
run r04_bench_c4_sparse_keys_unordered_output --workload c4 --c4-sparse --c4-unordered --steps 3 --warmup 1 --no-traffic --no-cpu-baseline
R=$GRAFT_REPO_ROOT
cd /tmp
for wl in c4 c4s; do
  args="--workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic"
  [ $wl = c4s ] && args="--workload c4 --c4-sparse --steps 2 --warmup 1 --no-cpu-baseline --no-traffic"
  rm -rf $R/$O/prof_$wl
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$wl/trace -- python $R/bench.py $args > $R/$O/prof_${wl}_trace.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_$wl/fetch -- python $R/bench.py $args > $R/$O/prof_${wl}_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_$wl/write -- python $R/bench.py $args > $R/$O/prof_${wl}_write.log 2>&1
  cd $R
  for d in trace fetch write; do echo "## pass: $d"; python tools/rocprof_summary.py $O/prof_$wl/$d; done > $O/r04_${wl}_rocprofv3_summary.md 2>&1
  find $O/prof_$wl -name "*.csv" -size +5M -delete
  cd /tmp
done
