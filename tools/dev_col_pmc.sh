#!/bin/bash
# development: SQ counters of the column path's kernels (one stream): tools/dev_col_pmc.sh TAG CONFIG
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SBO_COL_OVERLAP=${3:-0}
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$tag -- python3 $R/bench.py --config ${2:-H} --steps 3 --warmup 2 --cpu-sample 0 --no-extra > $OUT/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name=r["Kernel_Name"].split("(")[0].replace("void ","").replace("sbo::","")
        if name.startswith("k_col") or name.startswith("k_bpost"):
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name,d in sorted(agg.items()):
    print(name, " ".join(f"{k}={sum(v)/len(v):.3g}" for k,v in sorted(d.items())))
PY
