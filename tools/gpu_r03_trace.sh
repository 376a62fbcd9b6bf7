#!/bin/bash
# kernel trace of a few sweeps with given options: tools/gpu_r03_trace.sh TAG CONFIG "options"
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SBO_BENCH_OPTIONS="$3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config $2 --steps 6 --warmup 3 --cpu-sample 0 --no-extra > $OUT/trace.log 2>&1
echo "exit $?"
cd $R && python3 tools/timeline.py $1 > $OUT/timeline.txt 2>&1; cat $OUT/timeline.txt
