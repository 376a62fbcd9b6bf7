#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/dev_e_only.py 500000 4 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/pmc1 -- python3 $R/tools/dev_e_only.py 500000 1 > $OUT/pmc1.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc2 -- python3 $R/tools/dev_e_only.py 500000 1 > $OUT/pmc2.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("pmc1","pmc2"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "k_posterior" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0][-40:] + " lds=" + r["LDS_Block_Size"]+" vgpr="+r["VGPR_Count"]+"+"+r["Accum_VGPR_Count"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in agg.items():
            print(sub, k)
            for c, v in sorted(d.items()):
                print("    %-28s %s" % (c, ["%.4g" % x for x in v]))
PY
