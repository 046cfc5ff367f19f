set -u
O=gpurun_out/r06m; mkdir -p $O
for i in 1 2; do timeout 900 python -m pytest tests/test_apps_gpu.py -q -m gpu -x -s -k "mnist_meets" 2>&1 | grep -E "UMAP MNIST|passed|failed|assert|Error" | head -8; done | tee $O/mnist.txt
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_resnet_bf16_gpu.py -q -m gpu -x -k "conv or narrow or pair or stat" 2>&1 | tail -5 | tee $O/pytest_conv.txt
bash scripts/ab_libs.sh 3 lamp_amd/lib_base/liblamp_hip.so lamp_amd/lib/liblamp_hip.so 2>&1 | tee $O/ab.txt
AB_ARGS="--batch 256" bash scripts/ab_libs.sh 2 lamp_amd/lib_base/liblamp_hip.so lamp_amd/lib/liblamp_hip.so 2>&1 | tee $O/ab256.txt
hipcc -O3 --offload-arch=gfx950 scripts/microbench/gemm_kstep_loop.hip -o /tmp/gk.bin && /tmp/gk.bin 2>&1 | grep "mode [067]" | tee $O/mfma_rates.txt
bash scripts/prof_resnet.sh > $O/prof_head.txt 2>&1; cp gpurun_out/prof/timeline.txt $O/timeline.txt
