#!/bin/bash
# Round-3 evidence run on the GPU box: every bench line, then the rocprofv3 passes, then smoke().
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
ROUND=r03 sh tools/refresh_profiles.sh 2>&1 | tail -40
WLS="q1 q3 q3r c4" sh tools/run_prof_q1_q3.sh > gpurun_out/run_prof.log 2>&1
ls gpurun_out/*_rocprof_summary.md
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
