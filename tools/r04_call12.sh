#!/bin/bash
# Where does k_rp_aggregate (hashed fold, config 4 with sparse keys) spend its time?
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04w
mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
args="--workload c4 --c4-sparse --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-secondary"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python $R/bench.py $args > $R/$O/trace.log 2>&1
for c in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVES" "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_$n -- python $R/bench.py $args > $R/$O/pmc_$n.log 2>&1
done
cd $R
python tools/rocprof_summary.py $O/trace | head -30
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r04w/pmc_*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'].split('(')[0][:40]
            acc[k][row['Counter_Name']] += float(row['Counter_Value'])
        for k, v in acc.items():
            if 'k_rp_aggregate' in k or 'k_rp_scatter' in k:
                print(k, dict(v))
PY
for f in $O/pmc_*.log; do tail -2 $f; done
find $O -name "*.csv" -size +5M -delete
