set -u
O=gpurun_out/r05u; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "two_first_convolutions or input_gradient or convolution" 2>&1 | tail -5 > $O/pytest_a.txt
export LAMP_BENCH_ALSO=0
bash scripts/ab_env.sh LAMP_CONV_DGRAD_PAIR 0 1 3 > $O/ab_2048.txt 2>&1
