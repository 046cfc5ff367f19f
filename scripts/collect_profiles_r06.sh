#!/bin/bash
# copies what is judged from gpurun_out/refresh6/ (scripts/refresh_profiles_r06.sh) into profiles/r06_*
set -e
S=gpurun_out/refresh6; D=profiles
cp $S/pytest_gpu.txt $D/r06_pytest_gpu.txt
cp $S/bench.log $D/r06_resnet_step_bench.log
for n in resnet_step resnet_b256 resnet_b32 resnet_f32; do cp $S/${n}_kernel_stats.csv $D/r06_${n}_kernel_stats.csv; cp $S/${n}_last_step_breakdown.txt $D/r06_${n}_last_step_breakdown.txt; done
cp $S/resnet_step_timeline.txt $D/r06_resnet_step_timeline.txt
cp $S/class_rocprof.json $D/r06_class_rocprof.json
cp $S/pmc_traffic.json $D/r06_pmc_traffic.json
for w in gemm knn attention umap lm; do cp $S/${w}_kernel_stats.csv $D/r06_${w}_kernel_stats.csv; done
for w in gemm knn attention umap umap-e2e lm mlp epoch; do cp $S/${w}_bench.log $D/r06_${w}_bench.log; done
cp $S/gemm_ab.txt $D/r06_gemm_ab.txt
ls $D | grep r06 | wc -l
