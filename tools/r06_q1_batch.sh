#!/bin/bash
# k_agg_fast with the LDS reads of an iteration batched (slot map and first-row words of the four rows
# of a lane side by side, then the atomics; rare branches drained) against the previous library, alternating.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06q1b
mkdir -p $O
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline") or {}
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "kernel %.4f" % r.get("kernel_ms_per_step"), "frac %.3f" % r.get("frac"), "of columns ceiling", r.get("frac_of_q1_columns_ceiling"))
PY
}
for rep in 1 2 3; do
  for v in old new; do
    L=$GRAFT_REPO_ROOT/velox_amd/libvx355.so
    [ $v = old ] && L=$GRAFT_REPO_ROOT/velox_amd/variants/libvx355_old.so
    VX355_LIB_PATH=$L python bench.py --steps 30 --warmup 5 --no-traffic --no-cpu-baseline --no-secondary --detail $O/q1_${v}_$rep.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
    show $O/q1_${v}_$rep.json "q1 $v"
  done
done
for v in old new; do
  L=$GRAFT_REPO_ROOT/velox_amd/libvx355.so
  [ $v = old ] && L=$GRAFT_REPO_ROOT/velox_amd/variants/libvx355_old.so
  VX355_LIB_PATH=$L python bench.py --workload q1x4 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --no-secondary --detail $O/q1x4_$v.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
  show $O/q1x4_$v.json "q1x4 $v"
  VX355_LIB_PATH=$L python bench.py --workload c1 --steps 50 --warmup 5 --no-traffic --no-cpu-baseline --detail $O/c1_$v.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
  show $O/c1_$v.json "c1 $v"
done
python -m pytest tests/test_gpu_agg.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py tests/test_gpu_double_sums.py tests/test_gpu_bigint_sums.py tests/test_shim.py -q -m gpu -x 2>&1 | tail -3
