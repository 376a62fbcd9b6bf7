#!/bin/bash
# Quick kernel-trace of the default bench (run through gpurun from the repo root): per-kernel averages of one run.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --cpu-sample 0 > $OUT/bench.json 2> $OUT/bench.err && cat $OUT/bench.json &&
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $OUT/trace.log 2>&1
echo "exit $?"
