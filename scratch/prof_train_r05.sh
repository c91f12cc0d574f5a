#!/bin/bash
# steady-state kernel statistics + per-layer table of the bf16 training step (B=64, stages back to back): gpurun_out/<tag>/
tag=${1:-r05_mid}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --mode train --steps 3 --warmup 2 --train-batch 64 --train-storage bf16 --edge-products 1 --no-pipeline --layers > $out/bench_layers_train.json 2> $out/per_layer_table_train_bf16.txt
for K in 4 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_train_$K -o r -- python3 $R/bench.py --mode train --steps $K --warmup 2 --train-batch 64 --train-storage bf16 --edge-products 1 --no-pipeline > $out/stats_train_$K.log 2>&1
  cp $out/stats_train_$K/r_kernel_stats.csv $out/kernel_stats_train_bf16_steps$K.csv 2>/dev/null
  rm -rf $out/stats_train_$K
done
python3 $R/scratch/steady_stats.py $out/kernel_stats_train_bf16_steps4.csv $out/kernel_stats_train_bf16_steps0.csv 4 $out/kernel_stats_train_bf16_steady.csv > $out/kernel_stats_train_bf16_steady.txt
head -3 $out/kernel_stats_train_bf16_steady.txt | cut -c1-300
