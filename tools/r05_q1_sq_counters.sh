#!/bin/bash
# SQ counters of k_agg_fast on the TPC-H Q1 workload (separate rocprofv3 --pmc passes, kernel trace only).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_final/q1_sq
mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
args="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-traffic --detail ''"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_F64" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_LDS_ATOMIC_RETURN SQ_INSTS_SMEM"; do
  i=$((i+1))
  eval timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/p$i -- python $R/bench.py $args > $R/$O/p$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/r05_final/q1_sq/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_agg_fast" in r["Kernel_Name"]:
            t = tot[r["Counter_Name"]]; t[0] += float(r["Counter_Value"]); t[1] += 1
for k, (v, n) in sorted(tot.items()):
    print("%-28s %16.0f per launch (%d launches)" % (k, v / n, n))
PY
find $O -name "*.csv" -size +2M -delete
