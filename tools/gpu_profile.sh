#!/bin/bash
# rocprofv3 recipe (run through gpurun from the repo root): tools/gpu_profile.sh TAG CONFIG STEPS [full|-] ["extra bench args"]
#   trace/               rocprofv3 --kernel-trace --stats of the resident-model sweeps of CONFIG (--no-extra)
#   pmc_fetch|write|sq/  counter passes, each in its own run (no tracing domains next to --pmc)
#   bench.json           with "full": the default bench line (what the driver runs); else the --no-extra line of CONFIG
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
CFG=${2:-H}
STEPS=${3:-200}
EXTRA=${5:-}            # further bench.py arguments, e.g. "--sweep goose"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$4" = "full" ]; then
  python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
else
  python3 $R/bench.py --config $CFG --steps $STEPS --warmup 3 --cpu-sample 0 --no-extra $EXTRA > $OUT/bench.json 2> $OUT/bench.err || exit 1
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config $CFG --steps $STEPS --warmup 3 --cpu-sample 0 --no-extra $EXTRA > $OUT/trace.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --cpu-sample 0 --no-extra $EXTRA > $OUT/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --cpu-sample 0 --no-extra $EXTRA > $OUT/pmc_write.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --cpu-sample 0 --no-extra $EXTRA > $OUT/pmc_sq.log 2>&1
rc=$?
# the waits of the posterior kernel: LDS instructions and the cycles waves wait for them, VALU issue, the store path
[ $rc = 0 ] && rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --cpu-sample 0 --no-extra $EXTRA > $OUT/pmc_sq2.log 2>&1
echo "exit $rc $?"
