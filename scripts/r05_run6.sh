set -u
O=gpurun_out/r05g; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops_gpu.py -q -x -k "one_pass or mode_unique or sorting_family or batch_norm" 2>&1 | tail -15 > $O/pytest_bn.txt
cat $O/pytest_bn.txt
LAMP_BENCH_ALSO=0 bash scripts/ab_env.sh LAMP_BN_FUSED_FP 0 1 2 --dtype f32 2>&1 | tee $O/ab_bn_f32.txt
LAMP_BENCH_ALSO=0 bash scripts/ab_env.sh LAMP_BN_FUSED_FP 0 1 2 --dtype f64 2>&1 | tee $O/ab_bn_f64.txt
timeout 900 python -m pytest tests/test_autograd_gpu.py -q -x 2>&1 | tail -4 | tee $O/pytest_autograd.txt
