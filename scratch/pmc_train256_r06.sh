#!/bin/bash
# HBM traffic of the B=256 bf16 training step (the driver line's `train` leg) per kernel family: rocprofv3 --pmc passes (FETCH_SIZE and
# WRITE_SIZE separately), each for a run of 1 and of 3 timed steps behind one warm-up step.  Steady state = (3-step run - 1-step run) / 2:
# the plan build's zero fills of every buffer (torch.zeros) and the first step's calibration passes drop out.  `with_build` repeats the
# round-4 method (the 1-step run's total / 2 steps) for comparison with profiles/r04_pmc_traffic_train_b256.json.
# usage: bash scratch/pmc_train256_r05.sh [tag] [extra bench.py flags...]
R=$GRAFT_REPO_ROOT
tag=${1:-r06_t256}
shift
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for K in 1 3; do
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $out/pmct_${c}_$K -o r -- python3 $R/bench.py --mode train --steps $K --warmup 1 --train-batch 256 --train-storage bf16 --edge-products 1 --no-pipeline "$@" > $out/pmct_${c}_$K.log 2>&1
done
done
python3 - "$out" "$*" <<'PY'
import csv, glob, json, sys, collections
out, extra = sys.argv[1], sys.argv[2]
BF16 = ("conv3x3_bf16_kernel", "conv1x1_bf16_kernel", "conv1x1_bf16_multi_kernel", "wgrad3x3_bf16_kernel", "wgrad3x3_bf16_wide_kernel", "wgrad1x1_bf16_kernel", "conv_wgrad_wide_kernel", "conv_narrow_bf16_kernel")
EDGE = ("conv_f16x3_kernel", "conv_f16x3_big_kernel", "conv3x3_halo_f16_kernel", "conv1x1_f16x3_kernel", "conv1x1_ms_f16x3_kernel", "msblock_dil_kernel", "msdil_ps_kernel", "msdil1_kernel", "conv_f16_big1_kernel",
        "conv3x3_c4_f16_kernel", "conv3x3_rs_kernel", "conv3x3_rw_kernel", "bdcn_", "maxpool")
def fam(k):
    if any(s in k for s in BF16): return "bf16_conv"
    if any(s in k for s in EDGE): return "edge_net_frozen"
    return "rest"
def load(K):
    tot = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
    cnt = collections.Counter()
    for c in tot:
        for f in glob.glob(out + "/pmct_%s_%d/**/*counter_collection.csv" % (c, K), recursive=True):
            for r in csv.DictReader(open(f)):
                k = fam(r["Kernel_Name"])
                tot[c][k] += float(r["Counter_Value"])
                if c == "FETCH_SIZE": cnt[k] += 1
    return tot, cnt
t1, c1 = load(1)
t3, c3 = load(3)
res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python3 bench.py --mode train --steps K --warmup 1 --train-batch 256 "
                 "--train-storage bf16 --edge-products 1 --no-pipeline %s, K = 1 and K = 3" % extra,
       "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B/lane streaming reads), WRITE_SIZE as is, KB",
       "steady_state": {"what": "(3-step run - 1-step run) / 2 steps: per training step, plan build and first-step calibration excluded", "families": {}},
       "with_build": {"what": "1-step run / 2 profiled steps (round-4 method: includes the plan build's zero fills and the calibration passes)", "families": {}}}
for k in ("bf16_conv", "edge_net_frozen", "rest"):
    rd, wr = 2 * (t3["FETCH_SIZE"][k] - t1["FETCH_SIZE"][k]) * 1024 / 2, (t3["WRITE_SIZE"][k] - t1["WRITE_SIZE"][k]) * 1024 / 2
    res["steady_state"]["families"][k] = {"dispatches_per_step": (c3[k] - c1[k]) // 2, "hbm_read_gb_per_step": round(rd / 1e9, 2), "hbm_write_gb_per_step": round(wr / 1e9, 2),
                                          "hbm_bytes_per_launch": int((rd + wr) / max((c3[k] - c1[k]) // 2, 1))}
    rd, wr = 2 * t1["FETCH_SIZE"][k] * 1024 / 2, t1["WRITE_SIZE"][k] * 1024 / 2
    res["with_build"]["families"][k] = {"dispatches_per_step": c1[k] // 2, "hbm_read_gb_per_step": round(rd / 1e9, 2), "hbm_write_gb_per_step": round(wr / 1e9, 2)}
res["families"] = res["steady_state"]["families"]        # (what bench.py copies into the training line's roofline.traffic / step_hbm_gb)
for m in ("steady_state", "with_build"):
    res[m]["hbm_gb_per_step"] = round(sum(v["hbm_read_gb_per_step"] + v["hbm_write_gb_per_step"] for v in res[m]["families"].values()), 1)
import subprocess, os
res["sources_sha16"] = subprocess.run(["python3", "-c", "import importlib.util as u; s=u.spec_from_file_location('b', '%s/bench.py'); m=u.module_from_spec(s); s.loader.exec_module(m); print(m._sources_sha16())" % os.environ["GRAFT_REPO_ROOT"]], capture_output=True, text=True).stdout.strip()
res["edge_products"] = 1
json.dump(res, open(out + "/pmc_traffic_train_b256.json", "w"), indent=1)
print(json.dumps(res["steady_state"])); print(json.dumps(res["with_build"]))
PY
rm -rf $out/pmct_FETCH_SIZE_1 $out/pmct_WRITE_SIZE_1 $out/pmct_FETCH_SIZE_3 $out/pmct_WRITE_SIZE_3
