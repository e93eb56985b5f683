#!/bin/bash
# Round 6: operator_test repeated (a hang was seen once), then the whole GPU suite.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06suite
g++ -std=c++17 -g -Wall -I include tests/cpp/operator_test.cpp -L velox_amd -lvx355 -Wl,-rpath,$PWD/velox_amd -o /tmp/operator_test || exit 1
for i in 1 2 3 4 5 6 7 8; do
  timeout 100 /tmp/operator_test > gpurun_out/r06suite/optest_$i.out 2>&1 &
  PID=$!
  for s in $(seq 1 40); do sleep 1; kill -0 $PID 2>/dev/null || break; done
  if kill -0 $PID 2>/dev/null; then
    timeout 60 rocgdb -batch -p $PID -ex "thread apply all bt 25" > gpurun_out/r06suite/bt_$i.txt 2>&1
    kill -9 $PID
    echo "run $i HUNG"
  fi
  wait $PID; echo "run $i rc=$?"
done
python -m pytest tests -q -m gpu -x 2>&1 | tail -15
