#!/bin/bash
# development: kernel stats of the resident-model sweeps of CONFIG: tools/dev_col_stats.sh TAG CONFIG LEAN OVERLAP  (through gpurun)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SBO_BENCH_OPTIONS="col_overlap=${4:-1}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config ${2:-H} --lean ${3:-1} --steps 100 --warmup 3 --cpu-sample 0 --no-extra > $OUT/trace.log 2>&1
echo "lean ${3:-1} overlap ${4:-1} exit $?"
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$OUT/trace/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if float(r["TotalDurationNs"])>0 and int(r["Calls"])>=100:
        print(f'{r["Name"].split("(")[0].replace("void ","")[:50]:50s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
