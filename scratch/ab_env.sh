#!/bin/bash
# usage: bash scratch/ab_env.sh "<ENV=val ...>" "<ENV=val ...>" ...   one bench.py --mode infer run per setting ("" = defaults), value / ms printed
for s in "$@"; do
  env $s python3 bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/ab_one.json
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab_one.json').read()); print('%-60s value %8.2f  %7.3f ms' % (sys.argv[1] or '(defaults)', d['value'], d['ms_per_step']))" "$s"
done
