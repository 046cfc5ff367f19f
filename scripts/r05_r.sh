set -u
O=gpurun_out/r05r; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "accumulates_in_the_kernel or batch_norm" 2>&1 | tail -4 > $O/pytest_a.txt
timeout 600 python -m pytest tests/test_resnet_bf16_gpu.py tests/test_autograd_gpu.py -q -m gpu -x 2>&1 | tail -3 > $O/pytest_b.txt
export LAMP_BENCH_ALSO=0
for i in 1 2; do for B in 2048 256 32; do
  ms=$(python bench.py --no-cpu-baseline --batch $B 2>/dev/null | python3 -c "import json,sys; l=[x for x in sys.stdin if x.startswith('{')]; print(json.loads(l[-1])['ms_per_step'])")
  echo "B=$B ms_per_step $ms"
done; done > $O/lines.txt 2>&1
