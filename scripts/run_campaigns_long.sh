for c in "tests/fuzz_campaign.py --seconds 240" "tests/fuzz_campaign.py --views --seconds 240" "tests/fuzz_campaign_parser.py --seconds 180" "tests/fuzz_campaign_shards.py --seconds 90" "tests/fuzz_campaign_gzip.py --seconds 120"; do
  echo "== python $c"; python $c 2>&1 | grep -v amdgpu.ids | tail -2; echo "rc=$?"
done
