#!/bin/bash
# Runs on the GPU box (via gpurun): benches, rocprofv3 kernel stats and the PMC HBM-traffic passes of one round.
# usage: bash scratch/collect_profiles.sh <tag> [all|pmc]   -> gpurun_out/<tag>/...
tag=${1:-final}
only=${2:-all}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
if [ "$only" = all ]; then
python3 $R/bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python3 $R/bench.py --mode infer --steps 10 --warmup 3 --no-cpu-baseline --batch 128 > $out/bench_infer_b128.json 2>/dev/null
python3 $R/bench.py --mode train --steps 5 --warmup 2 --train-batch 256 --config baseline_adain_edge > $out/bench_train_adain.json 2>/dev/null
python3 $R/bench.py --mode infer --steps 10 --warmup 3 --no-cpu-baseline --chz 64 > $out/bench_infer_chz64.json 2>/dev/null
python3 $R/bench.py --mode infer --steps 5 --warmup 2 --no-cpu-baseline --layers > $out/bench_layers.json 2> $out/per_layer_table.txt
python3 $R/bench.py --mode train --steps 3 --warmup 2 --train-batch 64 --layers > $out/bench_layers_train.json 2> $out/per_layer_table_train.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o r -- python3 $R/bench.py --mode infer --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $out/stats.log 2>&1
cp $out/stats/r_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_train -o r -- python3 $R/bench.py --mode train --steps 3 --warmup 2 --train-batch 64 --no-pipeline > $out/stats_train.log 2>&1
cp $out/stats_train/r_kernel_stats.csv $out/kernel_stats_train.csv 2>/dev/null
fi
python3 $R/scratch/rs_dbg.py > $out/clock_stamps_rs_64to64.txt 2>&1
python3 $R/scratch/fused_dbg.py 32,32,32 > $out/clock_stamps_fused_96to32to32.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $out/pmc_$c -o r -- python3 $R/bench.py --mode infer --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > $out/pmc_$c.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $out/pmct_$c -o r -- python3 $R/bench.py --mode train --steps 1 --warmup 1 --train-batch 64 --no-pipeline > $out/pmct_$c.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, json, re, sys, collections
out = sys.argv[1]
SPLIT = ("conv_f16x3_kernel", "conv_f16x3_big_kernel", "conv3x3_halo_f16_kernel", "conv1x1_f16x3_kernel", "conv1x1_ms_f16x3_kernel",
         "fused_1x1_3x3_kernel", "msblock_dil_kernel", "conv3x3_c4_f16_kernel", "conv3x3_rs_kernel", "conv3x3_rw_kernel", "conv1x1_pool_f16x3_kernel", "conv3x3_wgrad_halo_f16_kernel")
FP32 = ("conv_igemm_kernel", "conv3x3_halo_kernel", "conv3x3_c4_kernel", "conv_wgrad", "conv3x3_wgrad_halo_kernel", "conv1x1_wgrad_allpairs_kernel")
fam = lambda k: "split_f16" if any(s in k for s in SPLIT) else ("fp32_conv" if any(s in k for s in FP32) else "other")
tot = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
per = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
cnt, pcnt = collections.Counter(), collections.Counter()
for c in tot:
    for f in glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = fam(r["Kernel_Name"])
            if "absmax_k" in r["Kernel_Name"]:      # calibration pass of the first run only (Plan._run_calibrating): not steady state
                continue
            m = re.search(r"(fused_1x1_3x3_kernel|msblock_dil_kernel|conv[a-z0-9_]*kernel|[a-z0-9_]+_k(?![a-z0-9_]))", r["Kernel_Name"])
            short = m.group(1) if m else r["Kernel_Name"].split("(")[0][-48:]
            tot[c][k] += float(r["Counter_Value"]); per[c][short] += float(r["Counter_Value"])
            if c == "FETCH_SIZE": cnt[k] += 1; pcnt[short] += 1
steps = 3
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), python3 bench.py --mode infer --steps 2 --warmup 1 --no-pipeline, B=64 inference",
       "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B/lane streaming reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is; counters are in KB",
       "steps_profiled": steps, "families": {}, "kernels": {}}
for k in ("split_f16", "fp32_conv", "other"):
    rd, wr, n = 2 * tot["FETCH_SIZE"][k] * 1024, tot["WRITE_SIZE"][k] * 1024, max(cnt[k], 1)
    res["families"][k] = {"dispatches_per_step": cnt[k] // steps, "hbm_read_gb_per_step": round(rd / steps / 1e9, 2),
                          "hbm_write_gb_per_step": round(wr / steps / 1e9, 2), "hbm_bytes_per_launch": int((rd + wr) / n)}
for k in sorted(per["FETCH_SIZE"], key=lambda k: -per["FETCH_SIZE"][k])[:16]:
    res["kernels"][k] = {"dispatches_per_step": pcnt[k] // steps, "hbm_read_gb_per_step": round(2 * per["FETCH_SIZE"][k] * 1024 / steps / 1e9, 2),
                         "hbm_write_gb_per_step": round(per["WRITE_SIZE"][k] * 1024 / steps / 1e9, 2)}
# training step (B=64, 2 steps profiled): totals only
tt = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = 0.0
    for f in glob.glob(out + "/pmct_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            v += float(r["Counter_Value"])
    tt[c] = v
res["train_b64"] = {"steps_profiled": 2, "hbm_read_gb_per_step": round(2 * tt["FETCH_SIZE"] * 1024 / 2 / 1e9, 2),
                    "hbm_write_gb_per_step": round(tt["WRITE_SIZE"] * 1024 / 2 / 1e9, 2)}
json.dump(res, open(out + "/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res["families"]))
PY
rm -rf $out/stats $out/stats_train $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmct_FETCH_SIZE $out/pmct_WRITE_SIZE
ls -la $out
