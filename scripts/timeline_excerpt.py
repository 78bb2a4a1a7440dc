"""Excerpt of a rocprofv3 --kernel-trace --memory-copy-trace run (CSV): the long rows of a window of the LAST burst of activity.
usage: python scripts/timeline_excerpt.py DIR FROM_MS TO_MS [MIN_US]"""
import csv, glob, sys
d, lo, hi = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 200.0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:40], r["Stream_Id"]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r["Direction"][12:], r["Stream_Id"]))
rows.sort()
start, mx = 0, rows[0][1]
for i in range(1, len(rows)):
    if rows[i][0] - mx > 30e6:
        start = i
    mx = max(mx, rows[i][1])
rows = rows[start:]
t0 = rows[0][0]
print(f"last burst: {len(rows)} rows, {(rows[-1][1] - t0) / 1e6:.1f} ms; rows of >= {min_us:.0f} us between {lo:.0f} and {hi:.0f} ms (start, duration, stream, what)")
for s, e, n, q in rows:
    t = (s - t0) / 1e6
    if lo < t < hi and (e - s) >= min_us * 1e3:
        print(f"{t:9.3f} +{(e - s) / 1e6:7.3f}  s{q:>3} {n}")
