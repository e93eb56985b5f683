#!/bin/bash
# End of round 6: the default line and the random-probe-order Q3 join again after k_pp_scatter_fast / k_pp_probe changed.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_final
mkdir -p $O
export TMPDIR=/tmp
( time python bench.py --detail $O/r06_bench_default.json ) > $O/r06_bench_default.stdout 2> $O/r06_bench_default.err
tail -c 8000 $O/r06_bench_default.stdout | tail -1 > $O/r06_bench_default.line.json
python -c "
import json; l=json.load(open('$O/r06_bench_default.line.json')); print('default line', len(json.dumps(l)), 'bytes', l['value'], l['ms_per_step'], l['roofline']['frac'], l['roofline'].get('frac_of_q1_columns_ceiling'), l['cpu_baseline']['value'])"
grep real $O/r06_bench_default.err
timeout 900 python bench.py --workload q3 --q3-random-probe --detail $O/r06_bench_q3_join_random_probe_order.json 2> $O/q3r.err | tail -1 > $O/r06_bench_q3_join_random_probe_order.line.json
python -c "
import json; d=json.load(open('$O/r06_bench_q3_join_random_probe_order.json')); r=d['roofline']; print('q3r', d['ms_per_step'], r['kernel'], r['kernel_ms_per_step'], r['frac'], r['traffic'], r['algorithmic_bytes_per_step'], {k: round(v,3) for k,v in d['kernels_ms_per_step'].items() if v > 0.05}, (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline_mt') or {}).get('value'))"
R=$GRAFT_REPO_ROOT
cd /tmp
wl=q3r
args="--workload q3 --q3-random-probe --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --detail ''"
rm -rf $R/$O/prof_$wl
eval timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$wl/trace -- python $R/bench.py $args > $R/$O/prof_${wl}_trace.log 2>&1
eval timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_$wl/fetch -- python $R/bench.py $args > $R/$O/prof_${wl}_fetch.log 2>&1
eval timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_$wl/write -- python $R/bench.py $args > $R/$O/prof_${wl}_write.log 2>&1
cd $R
for d in trace fetch write; do echo "## pass: $d"; python tools/rocprof_summary.py $O/prof_$wl/$d; done > $O/r06_${wl}_rocprofv3_summary.md 2>&1
find $O/prof_$wl -name "*.csv" -size +5M -delete
grep "k_pp_scatter\|k_pp_probe" $O/r06_${wl}_rocprofv3_summary.md | head -8
