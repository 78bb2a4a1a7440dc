"""Per-step timeline from a rocprofv3 --kernel-trace --output-format csv run: for the last steps of the run, each kernel's
start (relative to the step's first kernel), duration and the idle gap in front of it.  usage: python scripts/step_gaps.py DIR first_kernel_substring [steps]"""
import csv, glob, sys
d, first = sys.argv[1], sys.argv[2]
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("bzq::", "").replace("void ", "").split("(")[0][:44]))
rows.sort()
starts = [i for i, r in enumerate(rows) if first in r[2]]
if len(starts) < nsteps + 1:
    sys.exit("not enough steps")
gaps_tot, spans = [], []
for si in range(len(starts) - nsteps - 1, len(starts) - 1):
    a, b = starts[si], starts[si + 1]
    t0 = rows[a][0]
    prev_end = rows[a - 1][1] if a > 0 else t0
    busy = 0
    print(f"-- step: idle before first kernel {(t0 - prev_end) / 1e3:8.1f} us (host turnaround + sync)")
    pe = None
    for s, e, n in rows[a:b]:
        gap = (s - pe) / 1e3 if pe is not None else 0.0
        print(f"   {(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  {n}")
        pe = e; busy += e - s
    spans.append((rows[b - 1][1] - t0) / 1e3)
    print(f"   span {(rows[b - 1][1] - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, step period {(rows[b][0] - t0) / 1e3:.1f} us")
