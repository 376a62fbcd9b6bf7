#!/bin/bash
# kernel stats of the resident-model sweeps of a config: tools/gpu_kstats.sh TAG CONFIG [STEPS]   (run through gpurun from the repo root)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config ${2:-H} --steps ${3:-200} --warmup 3 --cpu-sample 0 --no-extra > $OUT/trace.log 2>&1
echo "exit $?"
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$OUT/trace/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if float(r["TotalDurationNs"])>0 and int(r["Calls"])>=100:
        print(f'{r["Name"].split("(")[0].replace("void ","")[:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
