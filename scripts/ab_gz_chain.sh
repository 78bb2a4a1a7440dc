# same-box A/B of the gzip ingest: chain kernels in LDS / through the L2 (beside a predecode), finder behind the copy or not,
# member checks deferred to the next call or not
cd $GRAFT_REPO_ROOT
run() {  # name, env...
  local name=$1; shift
  env "$@" python bench.py --ingest-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['ingest_mode']; print('$name', 'gzip', d['gzip']['value'], d['gzip']['ms'], 'first', d['gzip']['value_first_file_of_the_process'], '| plain', d['plain']['value'], 'bgzf', d['bgzf']['value'])"
}
for i in 1 2 3; do
  run "lds              " BZQ_GZ_CHAIN_L2=0 BZQ_GZ_EARLY_FIND=0 BZQ_GZ_DEFER=0
  run "l2  early        " BZQ_GZ_CHAIN_L2=1 BZQ_GZ_EARLY_FIND=1 BZQ_GZ_DEFER=0
  run "l2  early defer  " BZQ_GZ_CHAIN_L2=1 BZQ_GZ_EARLY_FIND=1 BZQ_GZ_DEFER=1
  run "l2        defer  " BZQ_GZ_CHAIN_L2=1 BZQ_GZ_EARLY_FIND=0 BZQ_GZ_DEFER=1
  run "lds early defer  " BZQ_GZ_CHAIN_L2=0 BZQ_GZ_EARLY_FIND=1 BZQ_GZ_DEFER=1
done
