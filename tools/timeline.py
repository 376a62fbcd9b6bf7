"""Print the kernel timeline of one bench step from a rocprofv3 kernel trace (gpurun_out/<tag>/trace/**/_kernel_trace.csv)."""
import csv, glob, sys
tag = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_bstage1<"
f = glob.glob(f"gpurun_out/{tag}/trace/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
s, e = idx[-2], idx[-1]
t0 = int(rows[s]["Start_Timestamp"]); prev = t0
for r in rows[s:e]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(st - t0) / 1e3:8.1f} {(en - st) / 1e3:7.1f} gap {(st - prev) / 1e3:6.1f}  grid {r['Grid_Size_X']:>9}x{r['Grid_Size_Y']:<6} {r['Kernel_Name'][:64]}")
    prev = en
