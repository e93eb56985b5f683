#!/bin/bash
# round 4, GPU call 7: the whole -m gpu suite
mkdir -p gpurun_out/r04g
cd "$GRAFT_REPO_ROOT"
( time timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r04g/tests_all.log 2>&1 ) 2> gpurun_out/r04g/tests_all.time
tail -15 gpurun_out/r04g/tests_all.log; tail -3 gpurun_out/r04g/tests_all.time
grep -n "^FAILED\|^ERROR" gpurun_out/r04g/tests_all.log | head -20
