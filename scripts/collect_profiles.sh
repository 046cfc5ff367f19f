#!/bin/bash
# Copies the outputs of scripts/refresh_profiles.sh (gpurun_out/refresh/, scratch) into profiles/ under the round's prefix.
set -eu
RND=${1:-r03}; O=gpurun_out/refresh; P=profiles
cp $O/bench.log $P/${RND}_resnet_step_bench.log
cp $O/kernel_stats.csv $P/${RND}_resnet_step_kernel_stats.csv
cp $O/last_step_breakdown.txt $P/${RND}_resnet_step_last_step_breakdown.txt
cp $O/pmc_traffic.json $P/${RND}_pmc_traffic.json
cp $O/pytest_gpu.txt $P/${RND}_pytest_gpu.txt
for w in gemm knn attention umap lm; do
  cp $O/bench_$w.log $P/${RND}_${w}_bench.log
  cp $O/kernel_stats_$w.csv $P/${RND}_${w}_kernel_stats.csv
done
for w in knn umap umap-e2e mlp; do [ -s $O/bench_full_$w.log ] && cp $O/bench_full_$w.log $P/${RND}_${w}_bench_with_cpu_baseline.log; done
for f in attention_probe umap_probe gemm_ab tf_ops_probe lm_kernels gemm_clock; do
  [ -s $O/$f.txt ] && cp $O/$f.txt $P/${RND}_$f.txt
done
ls $P | grep "^${RND}_"
