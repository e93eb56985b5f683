#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
summ() {
python - "$1" <<'PY'
import json, sys
try:
    lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
    d = json.loads(lines[-1])
    print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.05})
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
B="python bench.py --workload c5 --rows 200000000 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic"
VX355_BENCH_TORCH_NCCL=1 VX355_C5_PRESTEP=torch VX355_C5_CHUNKS=1 timeout 300 $B > gpurun_out/c14_pre.json 2> gpurun_out/c14_pre.err; summ gpurun_out/c14_pre.json; tail -3 gpurun_out/c14_pre.err
VX355_C5_KEYONLY=1 VX355_C5_CHUNKS=1 timeout 300 $B > gpurun_out/c14_keyonly.json 2> gpurun_out/c14_keyonly.err; summ gpurun_out/c14_keyonly.json
for mode in lib torch; do
  for ctr in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $ctr | tr ' ' '_')
    extra=""; [ $mode = torch ] && extra="--exchange torch"
    VX355_C5_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/c14_pmc_${mode}_$tag -o out --output-format csv -- python bench.py --workload c5 --rows 200000000 --steps 1 --warmup 1 --no-cpu-baseline --no-traffic $extra > /dev/null 2> gpurun_out/c14_pmc_${mode}_$tag.err
    python - gpurun_out/c14_pmc_${mode}_$tag <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if any(x in k for x in ("k_join_probe", "k_gather_deps", "k_join_insert", "k_emit")):
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in tot.items():
    print(d, k[:60], dict(v))
PY
    rm -rf gpurun_out/c14_pmc_${mode}_$tag
  done
done
