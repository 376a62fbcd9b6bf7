"""Developer helper: per-kernel durations of the last N dispatches in a rocprofv3 SQLite result (default output format of this image)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(db.execute("select name, start, end from kernels order by start"))
for name, s, e in rows[-n:]:
    print(f"{name[:72]:72s} {(e - s) / 1e3:10.1f} us")
