#!/bin/bash
# Re-measures what changed after tools/r05_final.sh ran (the hashed folds' partition groups): config 4 with sparse
# keys in both group-order forms, and its rocprofv3 passes. Same commands as r05_final.sh.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_final
mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; timeout 900 python bench.py "$@" --detail $O/$name.json 2> $O/$name.err | tail -1 > $O/$name.line.json; python -c "
import json; d=json.load(open('$O/$name.json')); print('$name', round(d['ms_per_step'],3), {k:v for k,v in d['kernels_ms_per_step'].items() if v>0.2})"; }
run r05_bench_c4_sparse_keys --workload c4 --c4-sparse --c4-unordered --steps 3 --warmup 1
run r05_bench_c4_sparse_keys_first_seen_order --workload c4 --c4-sparse --steps 3 --warmup 1 --no-cpu-baseline
R=$GRAFT_REPO_ROOT
cd /tmp
wl=c4s
args="--workload c4 --c4-sparse --c4-unordered --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --detail ''"
eval timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$wl/trace -- python $R/bench.py $args > $R/$O/prof_${wl}_trace.log 2>&1
eval timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_$wl/fetch -- python $R/bench.py $args > $R/$O/prof_${wl}_fetch.log 2>&1
eval timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_$wl/write -- python $R/bench.py $args > $R/$O/prof_${wl}_write.log 2>&1
cd $R
for d in trace fetch write; do echo "## pass: $d"; python tools/rocprof_summary.py $O/prof_$wl/$d; done > $O/r05_${wl}_rocprofv3_summary.md 2>&1
find $O/prof_$wl -name "*.csv" -size +5M -delete
