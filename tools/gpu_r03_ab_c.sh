#!/bin/bash
# A/B of sweep options on config C (SafeOpt and GoOSE): tools/gpu_r03_ab_c.sh OUTDIR "opt1" "opt2" ...
out=$1; shift
mkdir -p $out
i=0
for o in "$@"; do
  for sw in safeopt goose; do
    SBO_BENCH_OPTIONS="$o" timeout -k 10 300 python bench.py --config C --sweep $sw --steps 200 --warmup 20 --no-extra --cpu-sample 0 > $out/C_${sw}_$i.json 2> $out/C_${sw}_$i.err || exit 1
    python tools/print_bench.py $out/C_${sw}_$i.json "C $sw [$o]" 2>/dev/null | head -1
  done
  i=$((i+1))
done
