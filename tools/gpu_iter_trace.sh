#!/bin/bash
# kernel + copy trace of model-change iterations: tools/gpu_iter_trace.sh TAG CONFIG
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 $R/tools/dev_iteration_trace.py $2 > $OUT/trace.log 2>&1
echo "exit $?"; tail -4 $OUT/trace.log
cd $R && python3 tools/timeline_all.py gpurun_out/$1/trace 2.2 > $OUT/timeline.txt 2>&1; tail -80 $OUT/timeline.txt
