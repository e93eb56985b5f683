#!/bin/bash
# operator_test crashed once (SIGSEGV) and hung once as the first GPU process of a fresh box: run it under
# rocgdb until it fails and keep the backtraces.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06crash; mkdir -p $O
g++ -std=c++17 -g -O0 -Wall -I include tests/cpp/operator_test.cpp -L velox_amd -lvx355 -Wl,-rpath,$PWD/velox_amd -o /tmp/operator_test || exit 1
N=${1:-30}
bad=0
for i in $(seq 1 $N); do
  timeout 150 rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex "thread apply all bt 40" /tmp/operator_test > $O/run_$i.txt 2>&1
  if grep -q "match the expected results" $O/run_$i.txt && ! grep -q "SIGSEGV\|SIGABRT\|SIGBUS" $O/run_$i.txt; then rm -f $O/run_$i.txt; else bad=$((bad+1)); echo "run $i FAILED"; grep -n "SIGSEGV\|SIGABRT\|received signal" $O/run_$i.txt | head -3; fi
done
echo "bad=$bad of $N"
# and plain runs, to see the rate without the debugger
fail=0
for i in $(seq 1 $N); do
  timeout 100 /tmp/operator_test > $O/plain_$i.txt 2>&1; rc=$?
  if [ $rc -ne 0 ]; then fail=$((fail+1)); echo "plain run $i rc=$rc"; else rm -f $O/plain_$i.txt; fi
done
echo "plain failures=$fail of $N"
