#!/bin/bash
# every randomized parity campaign for S seconds each (by hand, at the end of a round): scripts/run_campaigns.sh 90 > gpurun_out/campaigns.txt
S=${1:-90}
for c in "tests/fuzz_campaign.py --seconds $S" "tests/fuzz_campaign.py --views --seconds $S" "tests/fuzz_campaign_parser.py --seconds $S" \
         "tests/fuzz_campaign_shards.py --seconds $S" "tests/fuzz_campaign_fasta.py $S 1" "tests/fuzz_campaign_inflate.py --seconds $S" \
         "tests/fuzz_campaign_gzip.py --seconds $S" "tests/fuzz_campaign_fasta_shards.py --seconds $S"; do
  echo "== python $c"; python $c 2>&1 | tail -2; echo "rc=$?"
done
