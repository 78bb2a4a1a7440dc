"""Timeline of a rocprofv3 --kernel-trace --memory-copy-trace run (CSV output): for the LAST `window_ms` of activity, every kernel /
copy with start and duration, and how much of the window each kind of work covers.  usage: python scripts/timeline.py DIR [name-filter]"""
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:40], r.get("Stream_Id", r.get("Queue_Id", ""))))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy"))[:30], ""))
rows.sort()
if not rows:
    sys.exit("no trace rows under " + d)
# the last run of the file: the rows behind the last gap of > 30 ms
start = 0
for i in range(1, len(rows)):
    if rows[i][0] - max(r[1] for r in rows[max(0, i - 50):i]) > 30e6:
        start = i
rows = rows[start:]
t0 = rows[0][0]
print(f"{len(rows)} rows, span {(rows[-1][1] - t0) / 1e6:.1f} ms")
for s, e, name, q in rows[:400]:
    if (e - s) > 200e3 or name.startswith("C"):
        print(f"{(s - t0) / 1e6:9.3f} ms  +{(e - s) / 1e6:7.3f} ms  {name}  {q}")
