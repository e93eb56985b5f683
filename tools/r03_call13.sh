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
VX355_C5_CHUNKS=1 timeout 300 python bench.py --workload c5 --rows 200000000 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/c13_c5_lib.json 2> gpurun_out/c13_c5_lib.err; summ gpurun_out/c13_c5_lib.json
VX355_BENCH_TORCH_NCCL=1 VX355_C5_CHUNKS=1 timeout 300 python bench.py --workload c5 --rows 200000000 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/c13_c5_lib_nccl.json 2> gpurun_out/c13_c5_lib_nccl.err; summ gpurun_out/c13_c5_lib_nccl.json; tail -2 gpurun_out/c13_c5_lib_nccl.err
VX355_C5_CHUNKS=1 timeout 300 python bench.py --workload c5 --exchange torch --rows 200000000 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/c13_c5_torch.json 2> gpurun_out/c13_c5_torch.err; summ gpurun_out/c13_c5_torch.json
