#!/bin/bash
# A/B of two builds of libsafebo.so on one box: alternates libsafebo_old.so / libsafebo_new.so under bench.py (config in $1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
  for v in old new; do
    cp safe-bayesian-optimization_amd/libsafebo_$v.so safe-bayesian-optimization_amd/libsafebo.so
    python bench.py --config $1 --steps ${2:-100} --warmup 10 --cpu-sample 0 --no-extra > gpurun_out/ab_$v.json 2>/dev/null
    echo -n "$v: "; python tools/print_bench.py gpurun_out/ab_$v.json | cut -c1-150
  done
done
cp safe-bayesian-optimization_amd/libsafebo_new.so safe-bayesian-optimization_amd/libsafebo.so
