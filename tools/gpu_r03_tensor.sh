#!/bin/bash
# K1t check + kernel trace on config D (one gpurun call)
set -e
mkdir -p gpurun_out
SBO_DEBUG_TENSOR=1 timeout -k 10 400 python tools/dev_tensor.py > gpurun_out/tensor.log 2>&1
cd /tmp && export TMPDIR=/tmp
DEV_TENSOR_ONLY=128 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tensor_prof -o t -- python3 $GRAFT_REPO_ROOT/tools/dev_tensor.py > $GRAFT_REPO_ROOT/gpurun_out/tensor_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/db_kernels.py gpurun_out/tensor_prof/t_results.db 30
