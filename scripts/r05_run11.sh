set -u
O=gpurun_out/r05l; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "pair_launch_sees or convolution_pair or pack_cache" 2>&1 | tail -15 | tee $O/pytest_a.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "wgrad or deferred" 2>&1 | tail -4 | tee -a $O/pytest_a.txt
