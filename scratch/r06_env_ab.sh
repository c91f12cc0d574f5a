#!/bin/bash
# round 6: A/B of one environment switch on the bf16 training leg inside one gpurun call.  usage: scratch/r06_env_ab.sh VAR val0 val1 [bench args]
var=$1; v0=$2; v1=$3; shift 3
T="--mode train --train-storage bf16 --train-steps 4 --no-cpu-baseline $*"
for v in $v0 $v1 $v0 $v1; do
  env $var=$v python bench.py $T 2>/dev/null > gpurun_out/_ab.json || exit 1
  python - "$var" "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/_ab.json")); t = d["train"] if "train" in d else d
print(sys.argv[1], sys.argv[2], "value", t["value"], "ms", t["ms_per_step"])
PY
done
