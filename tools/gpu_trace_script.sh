#!/bin/bash
# rocprofv3 --kernel-trace --stats of a dev script (run through gpurun from the repo root): tools/gpu_trace_script.sh <tag> <script> [args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/"$@" > $OUT/run.log 2>&1
rc=$?
f=$(ls $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -n 1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv
tail -n 12 $OUT/run.log
exit $rc
