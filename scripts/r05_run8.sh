set -u
O=gpurun_out/r05i; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "convolution_pair" 2>&1 | tail -12 | tee $O/pytest_a.txt
timeout 900 python -m pytest tests/test_resnet_bf16_gpu.py -q -x 2>&1 | tail -4 | tee -a $O/pytest_a.txt
LAMP_BENCH_ALSO=0 bash scripts/ab_lib.sh lamp_amd/lib_a/liblamp_hip.so lamp_amd/lib/liblamp_hip.so 3 2>&1 | tee $O/ab_narrow_pair.txt
LAMP_BENCH_ALSO=0 bash scripts/ab_lib.sh lamp_amd/lib_a/liblamp_hip.so lamp_amd/lib/liblamp_hip.so 2 --batch 256 2>&1 | tee $O/ab_narrow_pair_b256.txt
