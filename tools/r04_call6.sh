#!/bin/bash
# round 4, GPU call 6: the whole -m gpu suite + unfused Q1 shape + q1x4 timeline
mkdir -p gpurun_out/r04f
cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04f/tests_all.log 2>&1 ) 2> gpurun_out/r04f/tests_all.time
tail -8 gpurun_out/r04f/tests_all.log; tail -3 gpurun_out/r04f/tests_all.time
VX355_LOG_SHAPES=1 VX355_JIT=sync timeout 300 python bench.py --workload q1 --unfused --no-secondary --steps 5 --warmup 2 --no-traffic --no-cpu-baseline > gpurun_out/r04f/bench_q1_unfused.json 2> gpurun_out/r04f/bench_q1_unfused.err
grep "plan shape\|compiling\|loaded" gpurun_out/r04f/bench_q1_unfused.err | sort | uniq -c | head
python -c "
import json; d=json.loads(open('gpurun_out/r04f/bench_q1_unfused.json').read().strip().splitlines()[-1]); print('q1 unfused', d['ms_per_step'], d['kernels_ms_per_step'])"
timeout 300 python bench.py --workload q1 --unfused --no-secondary --steps 5 --warmup 2 --no-traffic --no-cpu-baseline > gpurun_out/r04f/bench_q1_unfused_second.json 2> gpurun_out/r04f/bench_q1_unfused_second.err
python -c "
import json; d=json.loads(open('gpurun_out/r04f/bench_q1_unfused_second.json').read().strip().splitlines()[-1]); print('q1 unfused, second process, default JIT mode', d['ms_per_step'], d['kernels_ms_per_step'])"
timeout 300 python tools/host_timeline.py q1x4 10 > gpurun_out/r04f/timeline_q1x4.log 2>&1
tail -3 gpurun_out/r04f/timeline_q1x4.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
