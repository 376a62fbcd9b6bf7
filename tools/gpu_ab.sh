#!/bin/bash
# A/B of bench options through kernel traces: tools/gpu_ab.sh <tag> "<opts A>" "<opts B>" ...   (run through gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for o in "$@"; do
  OUT=$R/gpurun_out/${tag}_$i
  mkdir -p $OUT
  [ "$o" = "-" ] && o=""
  SBO_BENCH_OPTIONS="$o" python3 $R/bench.py --cpu-sample 0 > $OUT/bench.json 2> $OUT/bench.err || exit 1
  SBO_BENCH_OPTIONS="$o" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $OUT/trace.log 2>&1 || exit 1
  i=$((i+1))
done
echo done
