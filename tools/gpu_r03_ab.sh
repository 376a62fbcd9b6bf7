#!/bin/bash
# A/B of sweep options on configs B and H (resident-model sweeps, no extras): usage tools/gpu_r03_ab.sh OUTDIR "opt1" "opt2" ...
out=$1; shift
mkdir -p $out
for cfg in B H; do
  i=0
  for o in "$@"; do
    SBO_BENCH_OPTIONS="$o" timeout -k 10 300 python bench.py --config $cfg --steps 200 --warmup 20 --no-extra --cpu-sample 0 > $out/${cfg}_$i.json 2> $out/${cfg}_$i.err || exit 1
    python tools/print_bench.py $out/${cfg}_$i.json "$cfg [$o]" || true
    i=$((i+1))
  done
done
