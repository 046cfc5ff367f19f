#!/bin/bash
# the GPU test files that reach the convolution kernels, one after the other (what fits the GPU minutes left; the whole suite: scripts/r05_full.sh)
O=gpurun_out/partial; mkdir -p $O; : > $O/partial_suite.txt
for f in tests/test_resnet_bf16_gpu.py tests/test_ops_gpu.py tests/test_autograd_gpu.py tests/test_apps_gpu.py tests/test_autograd_fuzz_gpu.py tests/test_fuzz_gpu.py tests/test_host_staging.py tests/test_data.py tests/test_transformer.py tests/test_distributed.py; do
  echo "$f: $(python -m pytest $f -q -m gpu -x 2>&1 | tail -1)" | tee -a $O/partial_suite.txt
done
