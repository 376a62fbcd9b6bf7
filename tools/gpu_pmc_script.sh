#!/bin/bash
# rocprofv3 --pmc pass over a dev script (run through gpurun from the repo root): tools/gpu_pmc_script.sh <tag> "<counters>" <script> [args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
CNT="$2"
shift; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --output-format csv -d $OUT/pmc -- python3 $R/"$@" > $OUT/run.log 2>&1
rc=$?
f=$(ls $OUT/pmc/*/*counter_collection.csv 2>/dev/null | head -n 1)
[ -n "$f" ] && cp $f $OUT/counters.csv
tail -n 5 $OUT/run.log
exit $rc
