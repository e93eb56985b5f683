#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_agg.py -m gpu -q -x -k "open_addressing" > gpurun_out/c11_hashed_tests.log 2>&1; echo "rc=$?" >> gpurun_out/c11_hashed_tests.log; tail -8 gpurun_out/c11_hashed_tests.log
summ() {
python - "$1" <<'PY'
import json, sys
try:
    lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
    d = json.loads(lines[-1])
    print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.05}, "frac", d["roofline"].get("frac"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
timeout 600 python bench.py --workload c4 --c4-sparse --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/c11_c4_sparse.json 2> gpurun_out/c11_c4_sparse.err; summ gpurun_out/c11_c4_sparse.json
VX355_AGG_RADIX_SPARSE=0 timeout 600 python bench.py --workload c4 --c4-sparse --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/c11_c4_sparse_atomics.json 2> gpurun_out/c11_c4_sparse_atomics.err; summ gpurun_out/c11_c4_sparse_atomics.json
timeout 600 python -m pytest tests/test_gpu_full_size.py -m gpu -q -k "c4" > gpurun_out/c11_full.log 2>&1; tail -3 gpurun_out/c11_full.log
VX355_C5_CHUNKS=1 VX355_C5_LIBEXCHANGE=1 timeout 300 python bench.py --workload c5 --exchange torch --rows 200000000 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/c11_c5_libexchange.json 2> gpurun_out/c11_c5_libexchange.err; summ gpurun_out/c11_c5_libexchange.json
