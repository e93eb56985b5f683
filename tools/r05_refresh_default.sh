#!/bin/bash
# Re-runs the driver's command (bench.py, no flags) and the config-1 lines after the last changes of round 5
# (20 timed steps by default; config 1 stepping through five C-ABI calls). Same commands as tools/r05_final.sh.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_final
mkdir -p $O
export TMPDIR=/tmp
( time python bench.py --detail $O/r05_bench_default.json ) > $O/r05_bench_default.stdout 2> $O/r05_bench_default.err
tail -c 8000 $O/r05_bench_default.stdout | tail -1 > $O/r05_bench_default.line.json
python -c "
import json; l=json.load(open('$O/r05_bench_default.line.json')); print('default line', len(json.dumps(l)), 'bytes', l['value'], l['ms_per_step'], l['steps'], l['roofline']['frac'], l['cpu_baseline']['value']); print({k:(round(v['ms_per_step'],3), v.get('frac')) for k,v in l['secondary'].items()})"
grep real $O/r05_bench_default.err
run() { name=$1; shift; timeout 900 python bench.py "$@" --detail $O/$name.json 2> $O/$name.err | tail -1 > $O/$name.line.json; python -c "
import json; d=json.load(open('$O/$name.json')); print('$name', round(d['ms_per_step'],4), {k:v for k,v in d['kernels_ms_per_step'].items() if v>0.002})"; }
run r05_bench_c1 --workload c1 --steps 200 --warmup 10
VX355_C1_ROTATE=1 run r05_bench_c1_replayed_input_in_infinity_cache --workload c1 --steps 200 --warmup 10 --no-traffic --no-cpu-baseline
VX355_C1_LEAN=0 run r05_bench_c1_through_the_python_wrapper_classes --workload c1 --steps 200 --warmup 10 --no-traffic --no-cpu-baseline
