O=gpurun_out/r05ae; mkdir -p $O
timeout 900 python -m pytest tests/test_resnet_bf16_gpu.py tests/test_fuzz_gpu.py tests/test_autograd_gpu.py -q -m gpu 2>&1 | tail -6 > $O/pytest_b.txt
