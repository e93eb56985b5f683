#!/bin/bash
# Round 6, config 5: full GPU suite, the bench line, rocprofv3 kernel trace + PMC passes.
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
mkdir -p gpurun_out/r06
python bench.py --workload c5 --rows 200000000 --steps 10 --warmup 3 --detail gpurun_out/r06/bench_c5_one_gpu.json 2>&1 | tail -1 > gpurun_out/r06/bench_c5_line.json
cat gpurun_out/r06/bench_c5_line.json
VX355_C5_CHUNKS=4 python bench.py --workload c5 --rows 200000000 --steps 5 --warmup 2 --no-traffic --no-cpu-baseline --detail gpurun_out/r06/bench_c5_one_gpu_4_chunks.json > /dev/null 2>&1
VX355_JOIN_REGROUP=0 python bench.py --workload c5 --rows 200000000 --steps 5 --warmup 2 --no-traffic --no-cpu-baseline --detail gpurun_out/r06/bench_c5_one_gpu_probe_not_regrouped.json > /dev/null 2>&1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06/c5_trace -- python $R/bench.py --workload c5 --rows 200000000 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-traffic --detail '' > $R/gpurun_out/r06/c5_trace.log 2>&1
cd $R
python tools/rocprof_summary.py gpurun_out/r06/c5_trace > gpurun_out/r06/c5_rocprofv3_summary.md 2>&1
bash tools/r06_c5_counters.sh final > gpurun_out/r06/c5_counters.txt 2>&1
find gpurun_out/r06 -name "*.csv" -size +1M -delete
tail -5 gpurun_out/r06/c5_rocprofv3_summary.md
