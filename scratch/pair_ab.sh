set -e
for v in 1 0 1 0; do
  EGNE_ZERO_AHEAD=$v python bench.py --mode train --train-storage bf16 --train-batch 64 --train-steps 12 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('zero_ahead=$v', d.get('value'), d.get('ms_per_step'))"
done
