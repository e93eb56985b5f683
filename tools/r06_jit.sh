#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
# fresh box = empty instance cache: the C++ program first (the case that crashed), twice from a cold cache
g++ -std=c++17 -g -O0 -Wall -I include tests/cpp/operator_test.cpp -L velox_amd -lvx355 -Wl,-rpath,$PWD/velox_amd -o /tmp/operator_test || exit 1
for i in 1 2 3 4 5 6 7 8; do
  d=/tmp/vxcache_$i; mkdir -p $d
  VX355_CACHE_DIR=$d VX355_LOG_SHAPES=1 timeout 120 /tmp/operator_test > /tmp/ot_$i.out 2>&1; echo "cold cache run $i rc=$? $(grep -c 'compiling an instance' /tmp/ot_$i.out) compiles ($(grep -c 'in the background' /tmp/ot_$i.out) in the background)"
done
python -m pytest tests/test_gpu_agg.py -q -m gpu -x -k "hiprtc or async or instantiat or cache" 2>&1 | tail -5
python -m pytest tests/test_gpu_agg.py -q -m gpu -x -k "test_async_instantiation_does_not_stall_the_first_batches" 2>&1 | tail -3
