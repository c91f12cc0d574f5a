#!/bin/bash
# copies the collected round-6 profile set from gpurun_out/<tag> (collect_profiles_r06.sh, pmc_train256_r06.sh, pmc_kernels.sh) into profiles/r06_*
tag=${1:-r06d}
g=gpurun_out/$tag
cp $g/bench.json profiles/r06_bench.json
for f in bench_train_adain_bf16 bench_train_chz64_bf16 bench_train_b256_bf16_rccl1 bench_train_b256_bf16_rccl1_nopipe; do cp $g/$f.json profiles/r06_$f.json; done
for f in kernel_stats_infer_steps0.csv kernel_stats_infer_steps10.csv kernel_stats_steady.csv kernel_stats_steady.txt kernel_stats_train_adain_bf16_steady.csv kernel_stats_train_adain_bf16_steady.txt \
         kernel_stats_train_bf16_steady.csv kernel_stats_train_bf16_steady.txt kernel_stats_train_bf16_steps0.csv kernel_stats_train_bf16_steps10.csv kernel_stats_train_chz64_bf16_steady.csv \
         kernel_stats_train_chz64_bf16_steady.txt per_layer_table.txt per_layer_table_train_bf16.txt pmc_traffic.json; do cp $g/$f profiles/r06_$f; done
cp gpurun_out/${tag}_t256/pmc_traffic_train_b256.json profiles/r06_pmc_traffic_train_b256.json
for k in train infer train_chz64; do [ -f gpurun_out/${tag}_pk_$k.txt ] && cp gpurun_out/${tag}_pk_$k.txt profiles/r06_pmc_kernels_$k.txt; done
[ -f gpurun_out/mask_mismatch.jsonl ] && cp gpurun_out/mask_mismatch.jsonl profiles/r06_mask_mismatch.jsonl
[ -f gpurun_out/bf16_horizon.json ] && cp gpurun_out/bf16_horizon.json profiles/r06_bf16_horizon.json
ls -la profiles | grep r06 | wc -l
