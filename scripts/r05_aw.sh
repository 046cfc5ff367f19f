set -u
O=gpurun_out/r05aw; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "convolution_of_batch_norm or folding or conv" 2>&1 | tail -5 > $O/pytest_a.txt
timeout 900 python -m pytest tests/test_resnet_bf16_gpu.py tests/test_autograd_gpu.py -q -m gpu -x 2>&1 | tail -5 > $O/pytest_b.txt
export LAMP_BENCH_ALSO=0
bash scripts/ab_env.sh LAMP_IG_FOLD_SMALL 0 1 3 --batch 256 > $O/ab_256.txt 2>&1
bash scripts/ab_env.sh LAMP_IG_FOLD_SMALL 0 1 2 --batch 32 > $O/ab_32.txt 2>&1
