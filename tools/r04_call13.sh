#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04m
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_agg.py -x -q -m gpu -k "radix or dense_folds or sorted or optimistic" 2>&1 | tail -5
for wave in 1 0; do
VX355_AGG_FOLD_WAVE=$wave timeout 600 python bench.py --workload c4 --steps 3 --warmup 1 --no-traffic --no-cpu-baseline --no-secondary 2> $O/c4_wave$wave.err | grep '^{"metric"' > $O/c4_wave$wave.json
python - <<PY
import json
d = json.load(open("$O/c4_wave$wave.json"))
print("c4 wave=$wave", round(d["ms_per_step"], 3), {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.02}, d.get("result_check"))
PY
done
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "c4 or config_4 or billion" 2>&1 | tail -3
