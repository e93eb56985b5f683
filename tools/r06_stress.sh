#!/bin/bash
# Repeats the two C++ consumer programs (queued pages, callbacks, worker threads) to catch a rare hang:
# backtraces of all threads with rocgdb when a run exceeds 40 s.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06stress; mkdir -p $O
g++ -std=c++17 -g -Wall -I include tests/cpp/operator_test.cpp -L velox_amd -lvx355 -Wl,-rpath,$PWD/velox_amd -o /tmp/operator_test || exit 1
g++ -std=c++17 -g -Wall -I include -I shim -I tests/velox_api_stub shim/Vx355Adapter.cpp shim/Vx355JoinAdapter.cpp tests/cpp/shim_operator_test.cpp -L velox_amd -lvx355 -lpthread -Wl,-rpath,$PWD/velox_amd -o /tmp/shim_operator_test || exit 1
N=${1:-40}
hung=0
for prog in operator_test shim_operator_test; do
  for i in $(seq 1 $N); do
    timeout 120 /tmp/$prog > $O/${prog}_$i.out 2>&1 &
    PID=$!
    for s in $(seq 1 400); do sleep 0.1; kill -0 $PID 2>/dev/null || break; done
    if kill -0 $PID 2>/dev/null; then
      timeout 60 rocgdb -batch -p $PID -ex "thread apply all bt 30" > $O/bt_${prog}_$i.txt 2>&1
      kill -9 $PID; hung=$((hung+1)); echo "$prog run $i HUNG"
    fi
    wait $PID; rc=$?
    [ $rc -ne 0 ] && { echo "$prog run $i rc=$rc"; tail -3 $O/${prog}_$i.out; }
    [ $rc -eq 0 ] && rm -f $O/${prog}_$i.out
  done
  echo "$prog: $N runs done"
done
echo "hung=$hung"
