set -u
O=gpurun_out/r05j; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv" 2>&1 | tail -5 | tee $O/pytest_a.txt
timeout 900 python -m pytest tests/test_resnet_bf16_gpu.py tests/test_autograd_gpu.py -q -x 2>&1 | tail -4 | tee -a $O/pytest_a.txt
LAMP_BENCH_ALSO=0 bash scripts/ab_env.sh LAMP_IG_W8 0 1 3 --batch 256 2>&1 | tee $O/ab_w8_b256.txt
LAMP_BENCH_ALSO=0 bash scripts/ab_env.sh LAMP_IG_W8 0 1 2 --batch 32 2>&1 | tee $O/ab_w8_b32.txt
LAMP_BENCH_ALSO=0 bash scripts/ab_env.sh LAMP_IG_W8 0 1 2 --batch 512 2>&1 | tee $O/ab_w8_b512.txt
