#!/bin/bash
# usage: bash scratch/ab_fit.sh "<ENV=val ...>" ...   one bench.py --mode infer --fit run per setting
for s in "$@"; do
  env $s python3 bench.py --mode infer --fit --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/ab_fit_err.txt | grep "^{" > gpurun_out/ab_one.json || { tail -5 gpurun_out/ab_fit_err.txt; continue; }
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab_one.json').read()); print('%-40s value %8.2f  %7.3f ms   with_fit %s' % (sys.argv[1] or '(defaults)', d['value'], d['ms_per_step'], d.get('with_fit_value')))" "$s"
done
