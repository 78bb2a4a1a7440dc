cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_final; rm -rf $O; mkdir -p $O
( echo "== python scripts/bench_bgzf_inflate.py (level 6)"; timeout 300 python scripts/bench_bgzf_inflate.py 2>&1 | grep -v amdgpu
  echo "== python scripts/bench_bgzf_inflate.py --level 1"; timeout 300 python scripts/bench_bgzf_inflate.py --level 1 2>&1 | grep -v amdgpu
  echo "== python scripts/bench_ingest_bgzf.py (3 GB)"; timeout 400 python scripts/bench_ingest_bgzf.py 2>&1 | grep -v amdgpu
  echo "== python scripts/bench_ingest_bgzf.py --gb 16"; timeout 400 python scripts/bench_ingest_bgzf.py --gb 16 2>&1 | grep -v amdgpu | sed -n 1,3p ) > $O/inflate.txt 2>&1
( timeout 1500 python scripts/bench_gzip.py --gb 1.05 --levels 1,6,9 2>&1 | grep -v amdgpu
  echo; echo "== a 3 GB file"; timeout 600 python scripts/bench_gzip.py --gb 3 --levels 6,1 --kinds pigz,multi 2>&1 | grep -v amdgpu ) > $O/gzip.txt 2>&1
( for c in "tests/fuzz_campaign.py --seconds 150" "tests/fuzz_campaign.py --views --seconds 150" "tests/fuzz_campaign_parser.py --seconds 150" "tests/fuzz_campaign_shards.py --seconds 120" "tests/fuzz_campaign_fasta.py 150 1" "tests/fuzz_campaign_fasta_shards.py --seconds 120" "tests/fuzz_campaign_inflate.py --seconds 120" "tests/fuzz_campaign_gzip.py --seconds 240"; do echo "== python $c"; timeout 600 python $c 2>&1 | grep -v amdgpu | tail -2; done ) > $O/campaigns.txt 2>&1
tail -3 $O/inflate.txt; tail -3 $O/gzip.txt; cat $O/campaigns.txt
