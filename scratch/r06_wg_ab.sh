#!/bin/bash
# round 6: A/B of the quad-per-workgroup 3x3 weight gradient (EGNE_WGRAD3_WIDE) on the training legs, inside one gpurun call
set -o pipefail
T="--mode train --train-storage bf16 --train-steps 4 --no-cpu-baseline"
for v in 0 1; do
  EGNE_WGRAD3_WIDE=$v python bench.py $T > gpurun_out/r06_ab_wg${v}_chz32.json 2>> gpurun_out/r06_ab.err || exit 1
  EGNE_WGRAD3_WIDE=$v python bench.py $T --chz 64 > gpurun_out/r06_ab_wg${v}_chz64.json 2>> gpurun_out/r06_ab.err || exit 1
done
python - <<'PY'
import json
for chz in (32, 64):
    for v in (0, 1):
        d = json.load(open("gpurun_out/r06_ab_wg%d_chz%d.json" % (v, chz)))
        t = d["train"] if "train" in d else d
        bk = t["roofline"].get("by_kernel") or t.get("roofline_secondary", {}).get("by_kernel")
        print("chz", chz, "wide", v, "value", t["value"], "ms", t["ms_per_step"], {k: (x.get("gb_per_s"), x.get("tflops"), x["time_share"]) for k, x in bk.items()})
PY
