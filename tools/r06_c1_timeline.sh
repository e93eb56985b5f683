#!/bin/bash
# Kernel timeline of config-1 steps (rocprofv3 --kernel-trace): start offsets and durations inside a step.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06c1
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d /tmp/c1trace -- python $R/bench.py --workload c1 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --detail '' > $O/trace.log 2>&1
cp $(find /tmp/c1trace -name "*kernel_trace.csv" | head -1) $O/c1_kernel_trace.csv; f=$(find /tmp/c1trace -name '*kernel_trace.csv' | head -1)
python - "$f" 2>&1 <<'PY' | tee $O/c1_timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].split("(")[0][:40], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# steps start at k_init_state
idx = [i for i, k in enumerate(ks) if k[0].startswith("k_init_state")]
for s in idx[-4:-1]:
    e = idx[idx.index(s) + 1]
    t0 = ks[s][1]
    print("step: %.1f us from this k_init_state to the next" % ((ks[e][1] - t0) / 1e3))
    prev_end = t0
    for name, a, b in ks[s:e]:
        print("  %-42s start %7.1f  dur %6.1f  gap before %6.1f" % (name, (a - t0) / 1e3, (b - a) / 1e3, (a - prev_end) / 1e3))
        prev_end = b
PY
