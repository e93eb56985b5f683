#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_final
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_agg.py -x -q -m gpu -k "first_seen or radix or dense_folds or hashed_folds" 2>&1 | tail -2
run() { name=$1; shift; timeout 900 python bench.py "$@" 2> $O/$name.err | grep '^{"metric"' | head -1 > $O/$name.json; python - <<PY
import json
d = json.load(open("$O/$name.json"))
print("$name", round(d["ms_per_step"], 3), "ms", (d.get("roofline") or {}).get("traffic"))
for k, v in d.get("secondary", {}).items():
    r = v.get("roofline") or {}
    print("   ", k, round(v["ms_per_step"], 3), r.get("kernel"), r.get("frac") and round(r["frac"], 3), r.get("traffic"))
PY
}
run r04_bench_default --steps 20 --warmup 5
run r04_bench_c4 --workload c4 --steps 3 --warmup 1
run r04_bench_c4_sparse_keys --workload c4 --c4-sparse --steps 3 --warmup 1
