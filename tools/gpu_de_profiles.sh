#!/bin/bash
# rocprofv3 evidence for configs D (128^4 on one GPU, K1g) and E (scattered 6-D points, fp32, K1c): kernel stats + counters.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/D_trace -- python3 $R/tools/dev_config_d_full.py > $OUT/D_trace.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/D_pmc_sq -- python3 $R/tools/dev_config_d_full.py > $OUT/D_pmc_sq.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/D_pmc_fetch -- python3 $R/tools/dev_config_d_full.py > $OUT/D_pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/D_pmc_write -- python3 $R/tools/dev_config_d_full.py > $OUT/D_pmc_write.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/E_trace -- python3 $R/bench.py --config E --points 2000000 --steps 3 --warmup 1 --cpu-sample 0 > $OUT/E_trace.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/E_pmc_sq -- python3 $R/bench.py --config E --points 2000000 --steps 2 --warmup 1 --cpu-sample 0 > $OUT/E_pmc_sq.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/E_pmc_fetch -- python3 $R/bench.py --config E --points 2000000 --steps 2 --warmup 1 --cpu-sample 0 > $OUT/E_pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/E_pmc_write -- python3 $R/bench.py --config E --points 2000000 --steps 2 --warmup 1 --cpu-sample 0 > $OUT/E_pmc_write.log 2>&1
echo "exit $?"
