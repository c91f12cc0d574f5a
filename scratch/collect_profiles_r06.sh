#!/bin/bash
# Runs on the GPU box (via gpurun): benches, rocprofv3 kernel stats (steady state by subtraction) and the PMC HBM-traffic passes of round 6.
# usage: bash scratch/collect_profiles_r06.sh <tag> [bench|stats|pmc|all]   -> gpurun_out/<tag>/...
tag=${1:-r06}
only=${2:-all}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
if [ "$only" = all ] || [ "$only" = bench ]; then
python3 $R/bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo bench done
python3 $R/bench.py --mode train --steps 4 --warmup 2 --train-batch 256 --train-storage bf16 --force-dist 2>/dev/null | grep "^{" > $out/bench_train_b256_bf16_rccl1.json
python3 $R/bench.py --mode train --steps 4 --warmup 2 --train-batch 256 --train-storage bf16 --force-dist --no-pipeline 2>/dev/null | grep "^{" > $out/bench_train_b256_bf16_rccl1_nopipe.json
python3 $R/bench.py --mode train --steps 4 --warmup 2 --train-batch 256 --config baseline_adain_edge --train-storage bf16 > $out/bench_train_adain_bf16.json 2>/dev/null; echo adain done
python3 $R/bench.py --mode train --steps 4 --warmup 2 --train-batch 256 --chz 64 --train-storage bf16 > $out/bench_train_chz64_bf16.json 2>/dev/null
python3 $R/bench.py --mode infer --steps 10 --warmup 3 --no-cpu-baseline --no-pipeline --layers > $out/bench_layers.json 2> $out/per_layer_table.txt
python3 $R/bench.py --mode train --steps 3 --warmup 2 --train-batch 64 --train-storage bf16 --edge-products 1 --no-pipeline --layers > $out/bench_layers_train.json 2> $out/per_layer_table_train_bf16.txt
fi
if [ "$only" = all ] || [ "$only" = stats ]; then
for K in 10 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$K -o r -- python3 $R/bench.py --mode infer --steps $K --warmup 3 --no-cpu-baseline --no-pipeline > $out/stats_$K.log 2>&1
  cp $out/stats_$K/r_kernel_stats.csv $out/kernel_stats_infer_steps$K.csv 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_train_$K -o r -- python3 $R/bench.py --mode train --steps $K --warmup 2 --train-batch 64 --train-storage bf16 --edge-products 1 --no-pipeline > $out/stats_train_$K.log 2>&1
  cp $out/stats_train_$K/r_kernel_stats.csv $out/kernel_stats_train_bf16_steps$K.csv 2>/dev/null
done
for cfg in "adain --config baseline_adain_edge" "chz64 --chz 64"; do
  set -- $cfg; tagc=$1; shift
  for K in 4 0; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_${tagc}_$K -o r -- python3 $R/bench.py --mode train --steps $K --warmup 2 --train-batch 64 --train-storage bf16 --edge-products 1 --no-pipeline "$@" > $out/stats_${tagc}_$K.log 2>&1
    cp $out/stats_${tagc}_$K/r_kernel_stats.csv $out/kernel_stats_train_${tagc}_bf16_steps$K.csv 2>/dev/null
    rm -rf $out/stats_${tagc}_$K
  done
  python3 $R/scratch/steady_stats.py $out/kernel_stats_train_${tagc}_bf16_steps4.csv $out/kernel_stats_train_${tagc}_bf16_steps0.csv 4 $out/kernel_stats_train_${tagc}_bf16_steady.csv > $out/kernel_stats_train_${tagc}_bf16_steady.txt
done
python3 $R/scratch/steady_stats.py $out/kernel_stats_infer_steps10.csv $out/kernel_stats_infer_steps0.csv 10 $out/kernel_stats_steady.csv > $out/kernel_stats_steady.txt
python3 $R/scratch/steady_stats.py $out/kernel_stats_train_bf16_steps10.csv $out/kernel_stats_train_bf16_steps0.csv 10 $out/kernel_stats_train_bf16_steady.csv > $out/kernel_stats_train_bf16_steady.txt
cat $out/kernel_stats_steady.txt $out/kernel_stats_train_bf16_steady.txt
fi
if [ "$only" = all ] || [ "$only" = pmc ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $out/pmc_$c -o r -- python3 $R/bench.py --mode infer --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > $out/pmc_$c.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $out/pmct_$c -o r -- python3 $R/bench.py --mode train --steps 1 --warmup 1 --train-batch 64 --train-storage bf16 --edge-products 1 --no-pipeline > $out/pmct_$c.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, json, re, sys, collections
out = sys.argv[1]
SPLIT = ("conv_f16x3_kernel", "conv_f16x3_big_kernel", "conv3x3_halo_f16_kernel", "conv1x1_f16x3_kernel", "conv1x1_ms_f16x3_kernel",
         "fused_1x1_3x3_kernel", "msblock_dil_kernel", "msdil_ps_kernel", "msdil1_kernel", "conv_f16_big1_kernel", "conv3x3_c4_f16_kernel", "conv3x3_rs_kernel", "conv3x3_rw_kernel", "conv1x1_pool_f16x3_kernel", "conv3x3_wgrad_halo_f16_kernel")
FP32 = ("conv3x3_narrow_f32_kernel", "conv_igemm_kernel", "conv3x3_halo_kernel", "conv3x3_c4_kernel", "conv_wgrad", "conv3x3_wgrad_halo_kernel", "conv1x1_wgrad_allpairs_kernel")
BF16 = ("conv3x3_bf16_kernel", "conv1x1_bf16_kernel", "conv1x1_bf16_multi_kernel", "wgrad3x3_bf16_kernel", "wgrad3x3_bf16_wide_kernel", "wgrad1x1_bf16_kernel", "conv_wgrad_wide_kernel", "conv_narrow_bf16_kernel")
def fam(k):
    if any(s in k for s in BF16): return "bf16_conv"
    if any(s in k for s in SPLIT): return "split_f16"
    if any(s in k for s in FP32): return "fp32_conv"
    return "other"
def collect(prefix, steps):
    tot = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
    per = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
    cnt, pcnt = collections.Counter(), collections.Counter()
    for c in tot:
        for f in glob.glob(out + "/%s_%s/**/*counter_collection.csv" % (prefix, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if "absmax_k" in r["Kernel_Name"] and prefix == "pmc":      # calibration pass of the first run only: not steady state
                    continue
                k = fam(r["Kernel_Name"])
                m = re.search(r"(fused_1x1_3x3_kernel|msblock_dil_kernel|msdil_ps_kernel|msdil1_kernel|conv[a-z0-9_]*kernel|wgrad[a-z0-9_]*kernel|[a-z0-9_]+_k(?![a-z0-9_]))", r["Kernel_Name"])
                short = m.group(1) if m else r["Kernel_Name"].split("(")[0][-48:]
                tot[c][k] += float(r["Counter_Value"]); per[c][short] += float(r["Counter_Value"])
                if c == "FETCH_SIZE": cnt[k] += 1; pcnt[short] += 1
    fams, kern = {}, {}
    for k in ("bf16_conv", "split_f16", "fp32_conv", "other"):
        rd, wr, n = 2 * tot["FETCH_SIZE"][k] * 1024, tot["WRITE_SIZE"][k] * 1024, max(cnt[k], 1)
        fams[k] = {"dispatches_per_step": cnt[k] // steps, "hbm_read_gb_per_step": round(rd / steps / 1e9, 2),
                   "hbm_write_gb_per_step": round(wr / steps / 1e9, 2), "hbm_bytes_per_launch": int((rd + wr) / n)}
    for k in sorted(per["FETCH_SIZE"], key=lambda k: -(2 * per["FETCH_SIZE"][k] + per["WRITE_SIZE"][k]))[:18]:
        kern[k] = {"dispatches_per_step": pcnt[k] // steps, "hbm_read_gb_per_step": round(2 * per["FETCH_SIZE"][k] * 1024 / steps / 1e9, 2),
                   "hbm_write_gb_per_step": round(per["WRITE_SIZE"][k] * 1024 / steps / 1e9, 2)}
    return fams, kern
fi, ki = collect("pmc", 3)
ft, kt = collect("pmct", 2)
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); inference: python3 bench.py --mode infer --steps 2 "
                 "--warmup 1 --no-pipeline (B=64, 3 steps profiled); training: --mode train --steps 1 --warmup 1 --train-batch 64 --train-storage bf16 "
                 "--no-pipeline (2 steps profiled, includes the frozen edge network's fp32 / split-f16 kernels)",
       "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B/lane streaming reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is; counters are in KB",
       "families": fi, "kernels": ki, "train_bf16_b64": {"families": ft, "kernels": kt,
       "hbm_read_gb_per_step": round(sum(v["hbm_read_gb_per_step"] for v in ft.values()), 2),
       "hbm_write_gb_per_step": round(sum(v["hbm_write_gb_per_step"] for v in ft.values()), 2)}}
import subprocess
res["sources_sha16"] = subprocess.run(["python3", "-c", "import importlib.util as u; s=u.spec_from_file_location('b', '%s/bench.py'); m=u.module_from_spec(s); s.loader.exec_module(m); print(m._sources_sha16())" % __import__("os").environ["GRAFT_REPO_ROOT"]], capture_output=True, text=True).stdout.strip()
json.dump(res, open(out + "/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res["families"])); print(json.dumps(res["train_bf16_b64"]["families"]))
PY
fi
rm -rf $out/stats_10 $out/stats_0 $out/stats_train_10 $out/stats_train_0 $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmct_FETCH_SIZE $out/pmct_WRITE_SIZE
ls -la $out
