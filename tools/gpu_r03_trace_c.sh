#!/bin/bash
# kernel trace of config C GoOSE sweeps
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --config C --sweep goose --steps 6 --warmup 3 --cpu-sample 0 --no-extra > $OUT/trace.log 2>&1
echo "exit $?"
cd $R && python3 tools/timeline_all.py gpurun_out/$1/trace 0.45 > $OUT/timeline.txt 2>&1; cat $OUT/timeline.txt
