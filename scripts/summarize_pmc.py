"""Per-dispatch PMC averages of the kernels whose name contains argv[2], from a rocprofv3 --pmc output directory."""
import glob
import os
import sqlite3
import sys

root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_")
for f in glob.glob(os.path.join(root, "**", "*.db"), recursive=True):
    db = sqlite3.connect(f)
    q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name order by kernel_name")
    cur = None
    for k, c, v, n in db.execute(q):
        if pat not in str(k):
            continue
        if k != cur:
            cur = k
            print(f"  {str(k).replace('bzq::', '').replace('void ', '').split('(')[0][:60]}  (dispatches {n})")
        print(f"      {c:24s} {v:18.1f}")
