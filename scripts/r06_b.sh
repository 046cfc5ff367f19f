set -u
O=gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_resnet_bf16_gpu.py -q -m gpu -x -k "loss_launch or plane_broadcast or tail_gradient or pooled or nll or batch_norm_pair" 2>&1 | tail -8 | tee $O/pytest_sel.txt
bash scripts/ab_libs.sh 3 lamp_amd/lib_base/liblamp_hip.so lamp_amd/lib/liblamp_hip.so 2>&1 | tee $O/ab.txt
AB_ARGS="--batch 256" bash scripts/ab_libs.sh 2 lamp_amd/lib_base/liblamp_hip.so lamp_amd/lib/liblamp_hip.so 2>&1 | tee $O/ab256.txt
