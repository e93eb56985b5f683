#!/bin/bash
# round 4, GPU call 2
mkdir -p gpurun_out/r04b
cd "$GRAFT_REPO_ROOT"
python -c "import torch; p=torch.cuda.get_device_properties(0); print('shared_memory_per_block', p.shared_memory_per_block, getattr(p,'shared_memory_per_block_optin',None), p.multi_processor_count)" > gpurun_out/r04b/props.log 2>&1
tail -1 gpurun_out/r04b/props.log
export VX355_LOG_SHAPES=1
timeout 900 python -m pytest tests/test_gpu_agg.py tests/test_gpu_async.py tests/test_gpu_bigint_sums.py tests/test_gpu_double_sums.py -x -q -m gpu > gpurun_out/r04b/tests_agg.log 2>&1
tail -5 gpurun_out/r04b/tests_agg.log
timeout 900 python -m pytest tests/test_gpu_dist_abi.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r04b/tests_dist_fuzz.log 2>&1
tail -5 gpurun_out/r04b/tests_dist_fuzz.log
timeout 600 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "q1" -s > gpurun_out/r04b/tests_full_q1.log 2>&1
tail -5 gpurun_out/r04b/tests_full_q1.log
timeout 300 python tools/host_timeline.py c1 50 > gpurun_out/r04b/timeline_c1.log 2>&1
tail -4 gpurun_out/r04b/timeline_c1.log
timeout 300 python tools/host_timeline.py q1x4 10 > gpurun_out/r04b/timeline_q1x4.log 2>&1
tail -4 gpurun_out/r04b/timeline_q1x4.log
timeout 300 python bench.py --workload q1x4 --steps 10 --warmup 3 --no-traffic --no-cpu-baseline > gpurun_out/r04b/bench_q1x4.json 2> gpurun_out/r04b/bench_q1x4.err
tail -c 900 gpurun_out/r04b/bench_q1x4.json; tail -3 gpurun_out/r04b/bench_q1x4.err
timeout 300 python bench.py --workload c1 --steps 50 --warmup 5 --no-traffic --no-cpu-baseline > gpurun_out/r04b/bench_c1.json 2> gpurun_out/r04b/bench_c1.err
tail -c 700 gpurun_out/r04b/bench_c1.json
VX355_C1_NULLS=0.5 timeout 300 python bench.py --workload c1 --steps 50 --warmup 5 --no-traffic --no-cpu-baseline > gpurun_out/r04b/bench_c1_halfnull.json 2>/dev/null
tail -c 700 gpurun_out/r04b/bench_c1_halfnull.json
timeout 300 python bench.py --workload c1 --rows 100000000 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline > gpurun_out/r04b/bench_c1_100m.json 2>/dev/null
tail -c 700 gpurun_out/r04b/bench_c1_100m.json
timeout 300 python bench.py --workload q1 --no-secondary --steps 10 --warmup 3 --no-traffic --no-cpu-baseline > gpurun_out/r04b/bench_q1.json 2>/dev/null
tail -c 500 gpurun_out/r04b/bench_q1.json
