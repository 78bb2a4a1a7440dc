# timeline of the gzip ingest (kernel + copy trace, CSV) under the environment given: scripts/trace_gz_ingest.sh TAG [ENV=VALUE ...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/gzt_$TAG; rm -rf $O; mkdir -p $O
env "$@" timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o t -- python scripts/sweep_ingest.py gzip 256:8 > $O/log.txt 2>&1
tail -1 $O/log.txt
python scripts/timeline_excerpt.py $O 90 150 80 > $O/excerpt.txt 2>&1
find $O -type f -size +4M -delete
