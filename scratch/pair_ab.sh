set -e
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_nets.py -x -q -m gpu 2>&1 | tail -3
for v in 1 1; do
  python bench.py --mode train --train-storage bf16 --train-batch 64 --train-steps 12 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('train B=64', d.get('value'), d.get('ms_per_step'))"
done
