#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 tools/scatter_bench.bin 500000000 > gpurun_out/c3_scatter_bench.txt 2>&1
cat gpurun_out/c3_scatter_bench.txt
