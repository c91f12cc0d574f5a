#!/bin/bash
# A/B of the overflow tests in the split-f16 epilogues: the same tree built with -DEGNE_NO_OVF_CHECK (scratch/novf_build) against the
# in-tree library, alternating, inference leg with and without the two-stage pipeline
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-ab_ovf}; mkdir -p $out
P=$R/scratch/novf_build/edge-guided-near-eye-image-analysis-for-head-mounted-displays_amd/csrc/libegne_hip.so
for i in 1 2 3; do
  EGNE_LIB=$P python3 $R/bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline > $out/novf_$i.json 2> $out/novf_$i.err
  python3 $R/bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline > $out/ovf_$i.json 2> $out/ovf_$i.err
done
python3 - "$out" <<'PY'
import json, sys
out = sys.argv[1]
for n in ("novf_1", "ovf_1", "novf_2", "ovf_2", "novf_3", "ovf_3"):
    try:
        d = json.loads(open("%s/%s.json" % (out, n)).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], "serial region", d["roofline"]["region_ms_per_step"])
    except Exception as e:
        print(n, "ERR", e)
PY
