#!/bin/bash
# A/B of environment settings through kernel traces: tools/gpu_env_ab.sh <tag> "VAR=1" "VAR=2" ... ("-" = none)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for o in "$@"; do
  OUT=$R/gpurun_out/${tag}_$i
  mkdir -p $OUT
  [ "$o" = "-" ] && o="SBO_NONE=1"
  export $o
  python3 $R/bench.py --cpu-sample 0 > $OUT/bench.json 2> $OUT/bench.err || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $OUT/trace.log 2>&1 || exit 1
  unset ${o%%=*}
  i=$((i+1))
done
echo done
