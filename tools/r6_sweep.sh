cd "${GRAFT_REPO_ROOT:-/root/repo}"
for cfg in "4 64" "4 48" "4 56" "5 48" "6 40" "3 85" "4 80" "4 64" "2 128" "4 40" "5 56" "6 48"; do
  set -- $cfg
  python bench.py --streams $1 --batch $2 --steps 100 --warmup 10 --no-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 x $2:', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms per step')"
done
