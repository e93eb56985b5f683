#!/bin/bash
cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/r04_final_tests.log 2>&1; tail -4 gpurun_out/r04_final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
