set -e
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for v in 0 1; do
  EGNE_ELREG_SIDE=$v python bench.py --no-cpu-baseline --train-storage bf16 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('elreg_side=$v', d.get('value'), d['config'].get('with_fit_value'), d['config'].get('latency_b2_ms_edge_seg_fit'), d['config'].get('train_value'), json.dumps(d.get('latency'))[:600])"
done
