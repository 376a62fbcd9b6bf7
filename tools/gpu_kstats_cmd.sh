#!/bin/bash
# kernel stats of an arbitrary python driver: tools/gpu_kstats_cmd.sh TAG script.py [args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/$@ > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log | head -1
cd $R && python3 - $OUT <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/trace/*/*_kernel_stats.csv'))[-1]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:80]:80s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}")
PY
