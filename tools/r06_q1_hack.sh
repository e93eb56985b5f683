#!/bin/bash
# Q1: k_agg_fast with the projection factors and the replica count as compile-time constants (a scratch
# build in velox_amd/variants/) against the product library, alternating on the same box; the seven-stream
# read ceiling of tools/q1_stream_bench.hip; sparse config 4 with 512 level-1 bins.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06q1h
mkdir -p $O
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline") or {}
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "kernel %s" % r.get("kernel_ms_per_step"), "frac %s" % r.get("frac"), "check", (d.get("result_check") or {}).get("ok"))
PY
}
for rep in 1 2 3; do
  for v in main hack; do
    L=$GRAFT_REPO_ROOT/velox_amd/libvx355.so
    [ $v = hack ] && L=$GRAFT_REPO_ROOT/velox_amd/variants/libvx355_q1hack.so
    VX355_LIB_PATH=$L python bench.py --steps 30 --warmup 5 --no-traffic --no-cpu-baseline --no-secondary --detail $O/q1_${v}_$rep.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
    show $O/q1_${v}_$rep.json "q1 $v"
  done
done
for v in main hack; do
  L=$GRAFT_REPO_ROOT/velox_amd/libvx355.so
  [ $v = hack ] && L=$GRAFT_REPO_ROOT/velox_amd/variants/libvx355_q1hack.so
  VX355_LIB_PATH=$L python bench.py --workload q1x4 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --no-secondary --detail $O/q1x4_$v.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
  show $O/q1x4_$v.json "q1x4 $v"
done
hipcc --offload-arch=gfx950 -O3 -o /tmp/q1_stream_bench tools/q1_stream_bench.hip 2>/dev/null && /tmp/q1_stream_bench | tee $O/q1_stream_bench.txt
for b in 0 512; do
  if [ $b = 0 ]; then unset VX355_AGG_RADIX_BINS; else export VX355_AGG_RADIX_BINS=$b; fi
  python bench.py --workload c4 --c4-sparse --c4-unordered --steps 3 --warmup 1 --no-traffic --no-cpu-baseline --detail $O/c4s_bins$b.json > /dev/null 2>$O/err.txt || tail -3 $O/err.txt
  show $O/c4s_bins$b.json "c4 sparse bins $b"
done
