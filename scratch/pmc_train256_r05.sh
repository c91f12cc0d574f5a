#!/bin/bash
# HBM traffic of the B=256 bf16 training step (the driver line's `train` leg) per kernel family: two rocprofv3 --pmc passes
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r05_t256
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $out/pmct_$c -o r -- python3 $R/bench.py --mode train --steps 1 --warmup 1 --train-batch 256 --train-storage bf16 --edge-products 1 --no-pipeline > $out/pmct_$c.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
BF16 = ("conv3x3_bf16_kernel", "conv1x1_bf16_kernel", "conv1x1_bf16_multi_kernel", "wgrad3x3_bf16_kernel", "wgrad1x1_bf16_kernel", "conv_wgrad_wide_kernel", "conv_narrow_bf16_kernel")
tot = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
cnt = collections.Counter()
for c in tot:
    for f in glob.glob(out + "/pmct_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = "bf16_conv" if any(s in r["Kernel_Name"] for s in BF16) else "rest"
            tot[c][k] += float(r["Counter_Value"])
            if c == "FETCH_SIZE": cnt[k] += 1
steps = 2
res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python3 bench.py --mode train --steps 1 --warmup 1 --train-batch 256 --train-storage bf16 --edge-products 1 --no-pipeline (2 steps profiled)",
       "correction": "FETCH_SIZE doubled, WRITE_SIZE as is, KB", "families": {}}
for k in ("bf16_conv", "rest"):
    rd, wr = 2 * tot["FETCH_SIZE"][k] * 1024, tot["WRITE_SIZE"][k] * 1024
    res["families"][k] = {"dispatches_per_step": cnt[k] // steps, "hbm_read_gb_per_step": round(rd / steps / 1e9, 2), "hbm_write_gb_per_step": round(wr / steps / 1e9, 2),
                          "hbm_bytes_per_launch": int((rd + wr) / max(cnt[k], 1))}
import subprocess, os
res["sources_sha16"] = subprocess.run(["python3", "-c", "import importlib.util as u; s=u.spec_from_file_location('b', '%s/bench.py'); m=u.module_from_spec(s); s.loader.exec_module(m); print(m._sources_sha16())" % os.environ["GRAFT_REPO_ROOT"]], capture_output=True, text=True).stdout.strip()
res["edge_products"] = 1
json.dump(res, open(out + "/pmc_traffic_train_b256.json", "w"), indent=1)
print(json.dumps(res["families"]))
PY
rm -rf $out/pmct_FETCH_SIZE $out/pmct_WRITE_SIZE
