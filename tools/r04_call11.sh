#!/bin/bash
mkdir -p gpurun_out/r04k
cd "$GRAFT_REPO_ROOT"
timeout 600 python tests/rccl_self_worker.py --with-torch > gpurun_out/r04k/rccl_self_torch.log 2>&1; echo "rc=$?"
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname" gpurun_out/r04k/rccl_self_torch.log | tail -12
VX355_COMM_FORCE_RCCL=1 VX355_C5_CHUNKS=4 timeout 600 python bench.py --workload c5 --rows 200000000 --steps 3 --warmup 1 --no-traffic --no-cpu-baseline > gpurun_out/r04k/bench_c5_rccl.json 2> gpurun_out/r04k/bench_c5_rccl.err
tail -3 gpurun_out/r04k/bench_c5_rccl.err
python -c "
import json; d=json.loads(open('gpurun_out/r04k/bench_c5_rccl.json').read().strip().splitlines()[-1]); print('c5 through rccl', d['ms_per_step'], d['config']['exchange'], d['result_check'], d['kernels_ms_per_step'])"
