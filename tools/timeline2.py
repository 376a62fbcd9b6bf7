#!/usr/bin/env python3
"""One sweep's kernel timeline from a rocprofv3 kernel trace: tools/timeline2.py gpurun_out/TAG [sweep index]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sbo::", "") for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("k_bstage1")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
i0, i1 = idx[k], idx[k + 1]
t0 = int(rows[i0]["Start_Timestamp"])
for r, n in zip(rows[i0:i1 + 1], names[i0:i1 + 1]):
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{n[:36]:36s} q{r.get('Queue_Id','?'):>3s} start {s/1e3:8.1f} end {e/1e3:8.1f} dur {(e-s)/1e3:7.1f}")
