for i in 1 2; do for v in 0 4096 131072 262144 163840; do
  python bench.py --no-cpu-baseline --no-extra-modes --views --ablate 1024 --opt force_dense=$v --steps 300 --warmup 5 --min-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fd=$v', d['value'], d['ms_per_step'], d['roofline_path']['ms']['aggregate'])"
done; done
