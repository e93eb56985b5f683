#!/bin/bash
# PMC counters of the config-5 kernels (separate rocprofv3 --pmc passes, kernel trace only).
# usage: tools/r06_c5_counters.sh <tag> [ENV=VALUE ...]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
O=gpurun_out/r06_c5_$tag
mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
args="--workload c5 --rows 200000000 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-traffic --detail ''"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "${EXTRA_SET:-GRBM_GUI_ACTIVE}"; do
  i=$((i+1))
  eval env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/p$i -- python $R/bench.py $args > $R/$O/p$i.log 2>&1
done
cd $R
python - "$O" <<'PY'
import csv, glob, collections, sys
tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").replace("vx::(anonymous namespace)::", "").split("(")[0].split("<")[0]
        if name.startswith("k_"):
            t = tot[name][r["Counter_Name"]]; t[0] += float(r["Counter_Value"]); t[1] += 1
for k in sorted(tot):
    if not any(x in k for x in ("k_grp", "k_join", "k_emit")):
        continue
    print(k)
    for c, (v, n) in sorted(tot[k].items()):
        print("   %-24s %18.0f per launch (%d launches)" % (c, v / n, n))
PY
find $O -name "*.csv" -size +1M -delete
