"""Print the merged kernel + memory-copy timeline of the last `span_ms` milliseconds of a rocprofv3 trace directory."""
import csv, glob, sys
d, span = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rows = []
for f in glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f"K grid {r['Grid_Size_X']:>8}x{r['Grid_Size_Y']:<5} {r['Kernel_Name'].split('(')[0][-70:]}"))
for f in glob.glob(f"{d}/**/*_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f"C {r.get('Direction', '')} {r.get('Size', '')} B"))
rows.sort()
tend = rows[-1][1]
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
sel = [r for r in rows if tend - (span + skip) * 1e6 <= r[0] <= tend - skip * 1e6]
t0, prev = sel[0][0], sel[0][0]
for st, en, what in sel:
    print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f} gap {(st - prev) / 1e3:7.1f}  {what}")
    prev = max(prev, en)
