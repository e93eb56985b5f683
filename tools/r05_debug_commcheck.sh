#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05i
export VX355_COMM_TRANSPORT=shm
rm -f /tmp/cc_id
for r in 0 1; do
  ( python -X faulthandler -m velox_amd.commcheck $r 2 0 /tmp/cc_id > gpurun_out/r05i/cc_rank$r.log 2>&1; echo "rank $r rc=$?" >> gpurun_out/r05i/cc_rank$r.log ) &
done
wait
tail -30 gpurun_out/r05i/cc_rank0.log
echo ----
tail -8 gpurun_out/r05i/cc_rank1.log
unset VX355_COMM_TRANSPORT
timeout 300 tools/partjoin_parts.bin > gpurun_out/r05i/partjoin_parts.txt 2>&1; cat gpurun_out/r05i/partjoin_parts.txt
