#!/bin/bash
# End of round 6: the default line and Q1's rocprofv3 passes again after the replica-per-lane layout.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_final
mkdir -p $O
export TMPDIR=/tmp
( time python bench.py --detail $O/r06_bench_default.json ) > $O/r06_bench_default.stdout 2> $O/r06_bench_default.err
tail -c 8000 $O/r06_bench_default.stdout | tail -1 > $O/r06_bench_default.line.json
python -c "
import json; l=json.load(open('$O/r06_bench_default.line.json')); print('default line', len(json.dumps(l)), 'bytes', l['value'], l['ms_per_step'], l['roofline']['frac'], l['roofline'].get('frac_of_q1_columns_ceiling'), l['cpu_baseline']['value'])"
grep real $O/r06_bench_default.err
timeout 900 python bench.py --workload q1x4 --steps 20 --warmup 5 --detail $O/r06_bench_q1x4.json 2> $O/q1x4.err | tail -1 > $O/r06_bench_q1x4.line.json
R=$GRAFT_REPO_ROOT
cd /tmp
for wl in q1 q1x4; do
  args="--steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-traffic --detail ''"
  [ $wl = q1x4 ] && args="--workload q1x4 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-traffic --detail ''"
  rm -rf $R/$O/prof_$wl
  eval timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$wl/trace -- python $R/bench.py $args > $R/$O/prof_${wl}_trace.log 2>&1
  eval timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_$wl/fetch -- python $R/bench.py $args > $R/$O/prof_${wl}_fetch.log 2>&1
  eval timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_$wl/write -- python $R/bench.py $args > $R/$O/prof_${wl}_write.log 2>&1
  cd $R
  for d in trace fetch write; do echo "## pass: $d"; python tools/rocprof_summary.py $O/prof_$wl/$d; done > $O/r06_${wl}_rocprofv3_summary.md 2>&1
  find $O/prof_$wl -name "*.csv" -size +5M -delete
  grep "k_agg_fast" $O/r06_${wl}_rocprofv3_summary.md | head -4 | cut -c1-200
  cd /tmp
done
