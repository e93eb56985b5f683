#!/bin/bash
# round 4, GPU call 4: parallel ingest, RCCL self path bisect, C1 timeline, streaming lines
mkdir -p gpurun_out/r04d
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_async.py tests/test_gpu_agg.py tests/test_gpu_join.py tests/test_cpp_consumer.py -x -q -m gpu > gpurun_out/r04d/tests.log 2>&1
tail -5 gpurun_out/r04d/tests.log
for steps in "counts all_gather" "all_gather" "columns" "columns all_gather_v" "all_gather_large" "edge"; do
  echo "== rccl self: $steps"
  timeout 300 python tests/rccl_self_worker.py $steps > gpurun_out/r04d/rccl_self_$(echo $steps | tr ' ' '_').log 2>&1
  echo "rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r04d/rccl_self_$(echo $steps | tr ' ' '_').log | tail -6
done
timeout 300 python tools/host_timeline.py c1 50 > gpurun_out/r04d/timeline_c1.log 2>&1
tail -3 gpurun_out/r04d/timeline_c1.log
VX355_AGG_SCRATCH_BLOCKS_PER_CU=1 timeout 300 python tools/host_timeline.py c1 50 > gpurun_out/r04d/timeline_c1_1bpc.log 2>&1
tail -3 gpurun_out/r04d/timeline_c1_1bpc.log
timeout 300 python bench.py --workload c1 --c1-stream --steps 10 --warmup 3 --no-traffic --no-cpu-baseline > gpurun_out/r04d/bench_c1_stream.json 2> gpurun_out/r04d/bench_c1_stream.err
python -c "
import json; d=json.loads(open('gpurun_out/r04d/bench_c1_stream.json').read().strip().splitlines()[-1]); print('c1 stream', d['ms_per_step'], d['workload_info'], d['kernels_ms_per_step'])"
tail -3 gpurun_out/r04d/bench_c1_stream.err
VX355_INGEST_THREADS=16 timeout 300 python bench.py --workload c1 --c1-stream --steps 10 --warmup 3 --no-traffic --no-cpu-baseline > gpurun_out/r04d/bench_c1_stream_16t.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/r04d/bench_c1_stream_16t.json').read().strip().splitlines()[-1]); print('c1 stream 16 threads', d['ms_per_step'], d['workload_info'])"
VX355_INGEST_PARALLEL=0 timeout 300 python bench.py --workload c1 --c1-stream --steps 10 --warmup 3 --no-traffic --no-cpu-baseline > gpurun_out/r04d/bench_c1_stream_serial.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/r04d/bench_c1_stream_serial.json').read().strip().splitlines()[-1]); print('c1 stream serial ingest', d['ms_per_step'], d['workload_info'])"
timeout 600 python bench.py --workload q1 --rows 60000000 --host-stream --steps 3 --warmup 1 --no-traffic --no-cpu-baseline --no-secondary > gpurun_out/r04d/bench_q1_stream.json 2> gpurun_out/r04d/bench_q1_stream.err
python -c "
import json; d=json.loads(open('gpurun_out/r04d/bench_q1_stream.json').read().strip().splitlines()[-1]); print('q1 stream', d['ms_per_step'], d['value'], d['workload_info'], d['kernels_ms_per_step'])"
tail -3 gpurun_out/r04d/bench_q1_stream.err
timeout 300 python bench.py --workload c4 --c4-sparse --steps 3 --warmup 1 --no-traffic --no-cpu-baseline > gpurun_out/r04d/bench_c4_sparse.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/r04d/bench_c4_sparse.json').read().strip().splitlines()[-1]); print('c4 sparse', d['ms_per_step'])"
