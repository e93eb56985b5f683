#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06suite2
( time python -m pytest tests -q -m gpu -x ) > gpurun_out/r06suite2/gpu_tests.log 2>&1; tail -4 gpurun_out/r06suite2/gpu_tests.log
g++ -std=c++17 -g -O0 -Wall -I include tests/cpp/operator_test.cpp -L velox_amd -lvx355 -Wl,-rpath,$PWD/velox_amd -o /tmp/operator_test || exit 1
bad=0
for i in $(seq 1 40); do
  d=/tmp/vxc_$i; mkdir -p $d
  VX355_CACHE_DIR=$d timeout 120 /tmp/operator_test > /tmp/ot_$i.out 2>&1 || { bad=$((bad+1)); echo "cold run $i rc=$?"; tail -2 /tmp/ot_$i.out; }
done
echo "cold-cache operator_test: $bad failures of 40"
bash tools/r06_stress.sh 15 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
