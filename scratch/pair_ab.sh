set -e
for v in 1 0 1 0; do
  EGNE_ELREG_SIDE=$v python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('elreg_side=$v', d.get('value'), d.get('ms_per_step'), d['roofline']['frac'], d['roofline']['region_ms_per_step'])"
done
