#!/bin/bash
# Where does tests/cpp/operator_test hang? Backtraces of all threads after 40 s.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06dbg
g++ -std=c++17 -g -Wall -I include tests/cpp/operator_test.cpp -L velox_amd -lvx355 -Wl,-rpath,$PWD/velox_amd -o /tmp/operator_test || exit 1
timeout 120 /tmp/operator_test > gpurun_out/r06dbg/optest.out 2>&1 &
PID=$!
sleep 45
if kill -0 $PID 2>/dev/null; then
  CH=$(pgrep -P $PID | head -1)
  timeout 60 rocgdb -batch -p ${CH:-$PID} -ex "thread apply all bt 25" > gpurun_out/r06dbg/bt.txt 2>&1
fi
wait $PID
echo "rc=$?" >> gpurun_out/r06dbg/optest.out
tail -5 gpurun_out/r06dbg/optest.out
grep -c Thread gpurun_out/r06dbg/bt.txt
python -m pytest tests -x -q -m gpu --deselect tests/test_cpp_consumer.py::test_cpp_operator_program_runs_on_the_gpu 2>&1 | tail -15
