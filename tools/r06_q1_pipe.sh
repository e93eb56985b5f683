#!/bin/bash
# Q1 / config 1: k_agg_fast with two rows per lane and the loads one iteration ahead (UNROLL 2, software
# pipelined) against four rows per lane in one register set (UNROLL 4) - VX355_AGG_FAST_UNROLL picks.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06q1p
mkdir -p $O
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline") or {}
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "kernel %s" % r.get("kernel_ms_per_step"), "frac %s" % r.get("frac"))
PY
}
for rep in 1 2; do
  for u in 4 2; do
    VX355_AGG_FAST_UNROLL=$u python bench.py --steps 30 --warmup 5 --no-traffic --no-cpu-baseline --no-secondary --detail $O/q1_u${u}_$rep.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
    show $O/q1_u${u}_$rep.json "q1 unroll $u"
  done
done
for u in 4 2; do
  VX355_AGG_FAST_UNROLL=$u python bench.py --workload q1x4 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --no-secondary --detail $O/q1x4_u$u.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
  show $O/q1x4_u$u.json "q1x4 unroll $u"
  VX355_AGG_FAST_UNROLL=$u python bench.py --workload c1 --steps 50 --warmup 5 --no-traffic --no-cpu-baseline --detail $O/c1_u$u.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
  show $O/c1_u$u.json "c1 unroll $u"
  python - $O/c1_u$u.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("   kernels", {k: round(v, 4) for k, v in d["kernels_ms_per_step"].items() if v > 0.002})
PY
done
VX355_AGG_FAST_UNROLL=2 python -m pytest tests/test_gpu_agg.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py tests/test_gpu_double_sums.py tests/test_gpu_bigint_sums.py -q -m gpu -x 2>&1 | tail -3
