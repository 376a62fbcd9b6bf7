#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("%-60s calls %4s avg %10.1f us  %5s%%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
