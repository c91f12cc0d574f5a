#!/bin/bash
# A/B of engine environment knobs on the B=64 inference step (pipelined, no fit / with fit)
run() { echo "== $*"; env "$@" python - <<'PY'
import sys, types
sys.path.insert(0, ".")
import bench
a = types.SimpleNamespace(gpus=1, batch=64, train_batch=256, config="baseline_edge", chz=32, no_pipeline=False, layers=False, fit=False)
bn = bench.Bench(a)
for fit in (False, True):
    B, dt, _ = bn.leg_infer(20, 5, fit=fit, events=False)
    print("fit=%s pipelined: %.3f ms/step" % (fit, 1e3 * dt / 20))
PY
}
run A=1
run EGNE_RW_MIN_W=60 EGNE_RW_MAX_COUTP=128
run EGNE_RW_MIN_W=60 EGNE_RW_MAX_COUTP=64
run EGNE_FIT_PRIO=1
