cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr; mkdir -p gpurun_out/tr
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tr -o t -- python scripts/bench_ingest_bgzf.py --gb 3 > gpurun_out/tr/log.txt 2>&1
python - <<'PY'
import sqlite3, glob
f = glob.glob('gpurun_out/tr/*.db')[0]
db = sqlite3.connect(f)
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'kernel' in t or 'copy' in t or 'memory' in t][:20])
rows = list(db.execute("select name, start, end from kernels where name like '%bgzf_inflate%' order by start"))
print(len(rows))
t0 = rows[0][1]
for n, s, e in rows[:40]:
    print(f"inflate start {(s-t0)/1e6:8.2f} ms dur {(e-s)/1e6:6.2f} ms")
PY
