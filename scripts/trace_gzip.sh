cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gz_kt; mkdir -p gpurun_out/gz_kt
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/gz_kt -o g -- python scripts/bench_gzip.py --gb 1.05 --levels 6 --kinds single --no-ingest > gpurun_out/gz_kt/log.txt 2>&1
python - <<'PY'
import sqlite3, glob
f = glob.glob('gpurun_out/gz_kt/*.db')[0]
db = sqlite3.connect(f)
print("== rocprofv3 --kernel-trace --stats -- python scripts/bench_gzip.py --gb 1.05 --levels 6 --kinds single --no-ingest   (3 decodes of a 0.52 GB single-member .gz = 1.0 GB of FASTQ each; 256 MiB pieces)")
for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 12"):
    print(f"  {str(r[0]).replace('bzq::','').replace('void ','').split('(')[0][:52]:52s} calls={r[1]:>5} total_us={float(r[2]):12.1f} avg_us={float(r[3]):10.2f} pct={float(r[4]):6.2f}")
PY
grep "decoder" gpurun_out/gz_kt/log.txt
