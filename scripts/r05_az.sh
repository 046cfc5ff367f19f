set -u
O=gpurun_out/r05az; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "conv or input_gradient" 2>&1 | tail -5 > $O/pytest_a.txt
timeout 900 python -m pytest tests/test_resnet_bf16_gpu.py tests/test_fuzz_gpu.py -q -m gpu -x 2>&1 | tail -5 > $O/pytest_b.txt
export LAMP_BENCH_ALSO=0
bash scripts/ab_lib.sh $PWD/lamp_amd/lib_prev/liblamp_hip.so $PWD/lamp_amd/lib/liblamp_hip.so 3 > $O/ab_2048.txt 2>&1
