set -u
O=gpurun_out/r06e; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_resnet_bf16_gpu.py -q -m gpu -x -k "loss_launch or plane_broadcast or tail_gradient or batch_norm_pair" 2>&1 | tail -8 | tee $O/pytest_sel.txt
bash scripts/ab_libs.sh 3 lamp_amd/lib_base/liblamp_hip.so lamp_amd/lib/liblamp_hip.so 2>&1 | tee $O/ab.txt
hipcc -O3 --offload-arch=gfx950 scripts/microbench/gemm_kstep_loop.hip -o /tmp/gk.bin && /tmp/gk.bin 2>&1 | tee $O/gemm_kstep_loop.txt
bash scripts/prof_resnet.sh > $O/prof_head.txt 2>&1; cp gpurun_out/prof/timeline.txt $O/timeline.txt; cp gpurun_out/prof/kernel_stats.csv $O/kernel_stats.csv
