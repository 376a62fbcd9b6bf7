#!/bin/bash
# kernel stats of resident-model sweeps of one config: tools/gpu_kstats.sh TAG CONFIG [SWEEP]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config ${2:-H} --sweep ${3:-safeopt} --steps ${4:-100} --warmup 3 --cpu-sample 0 --no-extra > $OUT/trace.log 2>&1
cd $R && python3 - $OUT <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/trace/*/*_kernel_stats.csv'))[-1]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:80]:80s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}")
PY
