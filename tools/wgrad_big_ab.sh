cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_bf16_gpu.py -x -q -m gpu -k "wgrad" 2>&1 | tail -2
for V in "1 2" "1 1" "0 2" "0 1" "1 2" "0 1"; do
set -- $V
LEOD_WGRAD_WIDE_BIG=$1 LEOD_WGRAD_WIDE_PF=$2 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-second-dtype 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('BIG=$1 PF=$2', d['ms_per_step'], 'frac', r['frac'], [(x['rows'], x['avg_us']) for x in r['by_rows']], 'wgrad fam', d['config']['family_ms_per_step'].get('leod_linear_wgrad'), d['config']['family_ms_per_step'].get('leod_linear_wgrad_gelu16'))"
done
