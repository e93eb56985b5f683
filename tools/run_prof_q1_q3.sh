# rocprofv3 kernel trace (--stats) + FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace only, as
# MI355X_MICROARCH.md prescribes) for the Q1 headline, the Q3 join in dbgen and in random probe order, and config 4.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for wl in ${WLS:-q1 q3 q3r c4}; do
  # the same step counts as the committed bench lines: a 4-step run is over before the clocks settle
  # (k_agg_fast 7.1 ms per step in a 3 + 1 step run, 6.4 ms in a 20 + 5 step run, with or without the tool)
  args="--steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-traffic"
  [ $wl = q3 ] && args="--workload q3 $args"
  [ $wl = q3r ] && args="--workload q3 --q3-random-probe $args"
  [ $wl = c4 ] && args="--workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$wl/trace -- python $R/bench.py $args > $R/gpurun_out/prof_${wl}_trace.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$wl/fetch -- python $R/bench.py $args > $R/gpurun_out/prof_${wl}_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$wl/write -- python $R/bench.py $args > $R/gpurun_out/prof_${wl}_write.log 2>&1
  cd $R
  for d in trace fetch write; do echo "## pass: $d"; python tools/rocprof_summary.py gpurun_out/prof_$wl/$d; done > gpurun_out/${wl}_rocprof_summary.md 2>&1
  find gpurun_out/prof_$wl -name "*.csv" -size +5M -delete
  find gpurun_out/prof_$wl -name "*.csv" | head -3
  cd /tmp
done
