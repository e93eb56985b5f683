#!/bin/bash
# rocprofv3 passes of config 1 at the round's final state (same commands as tools/r05_final.sh).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_final
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $O; rm -rf $O/prof_c1
cd /tmp
wl=c1
args="--workload c1 --steps 50 --warmup 5 --no-cpu-baseline --no-traffic --detail ''"
eval timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$wl/trace -- python $R/bench.py $args > $R/$O/prof_${wl}_trace.log 2>&1
eval timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_$wl/fetch -- python $R/bench.py $args > $R/$O/prof_${wl}_fetch.log 2>&1
eval timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_$wl/write -- python $R/bench.py $args > $R/$O/prof_${wl}_write.log 2>&1
cd $R
for d in trace fetch write; do echo "## pass: $d"; python tools/rocprof_summary.py $O/prof_$wl/$d; done > $O/r05_${wl}_rocprofv3_summary.md 2>&1
head -20 $O/r05_${wl}_rocprofv3_summary.md
