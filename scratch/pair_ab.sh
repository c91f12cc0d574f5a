set -e
for v in "1 1" "1 0" "0 0" "1 1" "1 0" "0 0"; do
  set -- $v
  EGNE_PAIR_BIAS=$1 EGNE_PAIR_BIAS_SIDE=$2 python bench.py --mode train --train-storage bf16 --train-batch 64 --train-steps 12 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('pair_bias=$1 side=$2', d.get('value'), d.get('ms_per_step'))"
done
