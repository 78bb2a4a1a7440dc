cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_final; mkdir -p $O
( echo "== python scripts/bench_gzip.py --gb 1.05 --levels 1,6,9 --kinds pigz,multi"; timeout 600 python scripts/bench_gzip.py --gb 1.05 --levels 1,6,9 --kinds pigz,multi 2>&1 | grep -v amdgpu
  echo; echo "== python scripts/bench_gzip.py --gb 1.05 --levels 6 --kinds single   (what gzip -6 writes; with the host's gzread beside it)"; timeout 600 python scripts/bench_gzip.py --gb 1.05 --levels 6 --kinds single 2>&1 | grep -v amdgpu
  echo; echo "== a 3 GB file: python scripts/bench_gzip.py --gb 3 --levels 6,1 --kinds pigz,multi"; timeout 600 python scripts/bench_gzip.py --gb 3 --levels 6,1 --kinds pigz,multi 2>&1 | grep -v amdgpu
  echo; echo "== a 12 GB file: python scripts/bench_gzip.py --gb 12 --levels 6 --kinds pigz"; timeout 600 python scripts/bench_gzip.py --gb 12 --levels 6 --kinds pigz 2>&1 | grep -v amdgpu
  echo; echo "== what a sequencer writes (quality runs, duplicate reads, poly-G tails: tests/gzip_util.py sequencer_like): python scripts/bench_gzip.py --data sequencer --gb 3 --levels 6,1 --kinds pigz"; timeout 600 python scripts/bench_gzip.py --data sequencer --gb 3 --levels 6,1 --kinds pigz 2>&1 | grep -v amdgpu ) > $O/gzip2.txt 2>&1
bash scripts/trace_gzip.sh > $O/gzip_kernels.txt 2>&1
( echo "== python scripts/bench_bgzf_inflate.py (level 6)"; timeout 300 python scripts/bench_bgzf_inflate.py 2>&1 | grep -v amdgpu
  echo "== python scripts/bench_bgzf_inflate.py --level 1"; timeout 300 python scripts/bench_bgzf_inflate.py --level 1 2>&1 | grep -v amdgpu
  echo "== python scripts/bench_ingest_bgzf.py (3 GB)"; timeout 400 python scripts/bench_ingest_bgzf.py 2>&1 | grep -v amdgpu | sed -n 1,3p
  echo "== python scripts/bench_ingest_bgzf.py --gb 16"; timeout 400 python scripts/bench_ingest_bgzf.py --gb 16 2>&1 | grep -v amdgpu | sed -n 1,3p ) > $O/inflate2.txt 2>&1
tail -4 $O/gzip2.txt; cat $O/gzip_kernels.txt | tail -16; tail -3 $O/inflate2.txt
