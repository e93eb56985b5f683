#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_memory_limit.py tests/test_gpu_kernels.py -q -m gpu -x -k "memory or limit or usage or presto or compress" 2>&1 | tail -25
