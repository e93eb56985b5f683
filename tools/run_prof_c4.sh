set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 600 python bench.py > gpurun_out/bench_q1.json 2> gpurun_out/bench_q1.err
tail -c 600 gpurun_out/bench_q1.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c4/trace -- python $R/bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_c4_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_c4/fetch -- python $R/bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_c4_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_c4/write -- python $R/bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_c4_write.log 2>&1
cd $R
for d in trace fetch write; do echo "## pass: $d"; python tools/rocprof_summary.py gpurun_out/prof_c4/$d; done > gpurun_out/c4_rocprof_summary.md 2>&1
# keep only the small CSVs out of the merge budget
find gpurun_out/prof_c4 -name "*.csv" -size +20M -delete
du -sh gpurun_out/prof_c4
head -50 gpurun_out/c4_rocprof_summary.md
