#!/usr/bin/env python3
"""Condense a gpurun_out/<tag> directory written by tools/gpu_profile.sh into profiles/<tag>_*.

Outputs: profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), profiles/<tag>_pmc.json
(per-kernel mean counter values per launch) and profiles/<tag>_bench.json (the bench line of the same run).
HBM traffic = FETCH_SIZE*1024 (x2 for wide 16-B/lane streaming reads on gfx950, MI355X_MICROARCH.md section HBM;
the x2 is NOT applied here because K1's reads are 8-B/lane fragment loads -- uncalibrated, reported raw) +
WRITE_SIZE*1024 (exact).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
ks = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if ks:
    shutil.copy(ks[-1], os.path.join(dst, f"{tag}_kernel_stats.csv"))      # (the newest run of this tag)
pmc = collections.defaultdict(dict)
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in sorted(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            pmc[name].setdefault("VGPR_Count", int(r["VGPR_Count"]))
            pmc[name].setdefault("Accum_VGPR_Count", int(r["Accum_VGPR_Count"]))
            pmc[name].setdefault("LDS_Block_Size", int(r["LDS_Block_Size"]))
        for name, d in agg.items():
            for cname, vals in d.items():
                pmc[name][cname] = sum(vals) / len(vals)
                pmc[name]["launches_" + cname] = len(vals)
for name, d in pmc.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch_raw"] = (d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
json.dump(pmc, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
b = os.path.join(src, "bench.json")
if os.path.exists(b):
    shutil.copy(b, os.path.join(dst, f"{tag}_bench.json"))
print("wrote", sorted(os.listdir(dst)))
