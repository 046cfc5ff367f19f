set -u
O=gpurun_out/r05full; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -rf 2>&1 | grep -E "passed|failed|error|^FAILED|^ERROR" | tail -12 | tee $O/pytest_gpu.txt
