#!/bin/bash
# round 6: A/B of the bf16 3x3 kernel's two-block form (EGNE_B3_MB) on the training legs; called inside one gpurun call
set -o pipefail
mkdir -p gpurun_out
T="--mode train --train-storage bf16 --train-steps 4 --no-cpu-baseline"
for mb in 1 2; do
  EGNE_B3_MB=$mb python bench.py $T > gpurun_out/r06_ab_b3_mb${mb}_chz32.json 2>> gpurun_out/r06_ab.err || exit 1
  echo "mb=$mb chz32 done"
  EGNE_B3_MB=$mb python bench.py $T --chz 64 > gpurun_out/r06_ab_b3_mb${mb}_chz64.json 2>> gpurun_out/r06_ab.err || exit 1
  echo "mb=$mb chz64 done"
done
python - <<'PY'
import json
for chz in (32, 64):
    for mb in (1, 2):
        d = json.load(open("gpurun_out/r06_ab_b3_mb%d_chz%d.json" % (mb, chz)))
        t = d["train"] if "train" in d else d
        bk = t["roofline"].get("by_kernel") or t.get("roofline_secondary", {}).get("by_kernel")
        print("chz", chz, "mb", mb, "value", t["value"], "ms", t["ms_per_step"], {k: (v.get("gb_per_s"), v.get("tflops"), v["time_share"]) for k, v in bk.items()})
PY
