#!/bin/bash
# round 4, GPU call 1: new 4-key shape + scratch flush: tests, timeline, bench lines
mkdir -p gpurun_out/r04a
cd "$GRAFT_REPO_ROOT"
export VX355_LOG_SHAPES=1
timeout 900 python -m pytest tests/test_gpu_agg.py -x -q -m gpu > gpurun_out/r04a/tests_agg.log 2>&1
tail -5 gpurun_out/r04a/tests_agg.log
timeout 600 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "four_keys" -s > gpurun_out/r04a/tests_full_4key.log 2>&1
tail -5 gpurun_out/r04a/tests_full_4key.log
timeout 300 python tools/host_timeline.py c1 50 > gpurun_out/r04a/timeline_c1.log 2>&1
tail -4 gpurun_out/r04a/timeline_c1.log
timeout 300 python bench.py --workload q1x4 --steps 10 --warmup 3 --no-traffic --no-cpu-baseline > gpurun_out/r04a/bench_q1x4.json 2> gpurun_out/r04a/bench_q1x4.err
tail -c 1500 gpurun_out/r04a/bench_q1x4.json; tail -3 gpurun_out/r04a/bench_q1x4.err
timeout 300 python bench.py --workload c1 --steps 50 --warmup 5 --no-traffic --no-cpu-baseline > gpurun_out/r04a/bench_c1.json 2> gpurun_out/r04a/bench_c1.err
tail -c 800 gpurun_out/r04a/bench_c1.json
VX355_AGG_SCRATCH_FLUSH=0 timeout 300 python bench.py --workload c1 --steps 50 --warmup 5 --no-traffic --no-cpu-baseline > gpurun_out/r04a/bench_c1_atomics.json 2>/dev/null
tail -c 800 gpurun_out/r04a/bench_c1_atomics.json
timeout 300 python bench.py --workload q1 --no-secondary --steps 10 --warmup 3 --no-traffic --no-cpu-baseline > gpurun_out/r04a/bench_q1.json 2>/dev/null
tail -c 800 gpurun_out/r04a/bench_q1.json
