set -e
for v in 1 0 1 0; do
  EGNE_WGRAD_SIDE=$v python bench.py --mode train --train-storage fp32 --train-batch 64 --train-steps 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('fp32 wgrad_side=$v', d.get('value'), d.get('ms_per_step'))"
done
for v in 1 0; do
  EGNE_WGRAD_SIDE=$v python bench.py --mode train --train-storage bf16 --train-batch 256 --train-steps 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('bf16 B=256 wgrad_side=$v', d.get('value'), d.get('ms_per_step'))"
done
