#!/bin/bash
# steady-state kernel statistics of the B=256 bf16 training step (no pipeline): runs of 3 and of 1 timed steps, difference / 2
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-qs256}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for K in 3 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$K -o r -- python3 $R/bench.py --mode train --steps $K --warmup 1 --train-batch 256 --train-storage bf16 --edge-products 1 --no-pipeline --no-cpu-baseline > $out/st_$K.log 2>&1
  cp $out/st_$K/r_kernel_stats.csv $out/kernel_stats_train256_steps$K.csv
  rm -rf $out/st_$K
done
python3 $R/scratch/steady_stats.py $out/kernel_stats_train256_steps3.csv $out/kernel_stats_train256_steps1.csv 2 $out/kernel_stats_train256_steady.csv > $out/kernel_stats_train256_steady.txt
head -5 $out/kernel_stats_train256_steady.txt | cut -c1-300
