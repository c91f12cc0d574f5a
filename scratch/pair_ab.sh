set -e
for v in 1 0 1 0; do
  EGNE_PIPE_EARLY_FREE=$v python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('early_free=$v', d.get('value'), d.get('ms_per_step'))"
  EGNE_PIPE_EARLY_FREE=$v python bench.py --mode infer --fit --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('early_free=$v with fit', d.get('value'), d.get('ms_per_step'))"
done
for v in 1 0; do
  EGNE_PIPE_EARLY_FREE=$v python bench.py --mode train --train-storage bf16 --train-batch 64 --train-steps 12 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('early_free=$v train', d.get('value'), d.get('ms_per_step'))"
done
