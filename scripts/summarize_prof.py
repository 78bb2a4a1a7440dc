"""Summarise rocprofv3 (rocpd sqlite) outputs: kernel stats + per-kernel PMC averages, as text."""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]


def short(name):
    name = str(name).replace("bzq::", "").replace("void ", "")
    return name.split("(")[0][:48]


for sub in ("kt", "pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    for f in glob.glob(os.path.join(root, sub, "*.db")):
        db = sqlite3.connect(f)
        if sub == "kt":
            print(f"== {sub}: rocprofv3 --kernel-trace --stats (top_kernels; durations in us)")
            for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 14"):
                print(f"  {short(r[0]):48s} calls={r[1]:>5} total_us={float(r[2]):12.1f} avg_us={float(r[3]):10.2f} pct={float(r[4]):6.2f}")
        else:
            print(f"== {sub}: per-dispatch counter averages")
            q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                 "group by kernel_name, counter_name order by kernel_name")
            cur = None
            for k, c, v, n in db.execute(q):
                if "k_" not in str(k):
                    continue
                if k != cur:
                    cur = k
                    print(f"  {short(k)}  (dispatches {n})")
                print(f"      {c:24s} {v:18.1f}")
