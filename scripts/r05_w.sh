set -u
O=gpurun_out/r05w; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "two_first_convolutions or input_gradient or convolution_pair" 2>&1 | tail -12 > $O/pytest_a.txt
timeout 900 python -m pytest tests/test_resnet_bf16_gpu.py tests/test_autograd_gpu.py -q -m gpu -x 2>&1 | tail -12 > $O/pytest_b.txt
export LAMP_BENCH_ALSO=0
for i in 1 2 3; do
  ms=$(python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; l=[x for x in sys.stdin if x.startswith('{')]; print(json.loads(l[-1])['ms_per_step'])")
  echo "new ms_per_step $ms"
  ms=$(LAMP_LIB_PATH=$PWD/lamp_amd/lib/liblamp_hip_prev.so python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; l=[x for x in sys.stdin if x.startswith('{')]; print(json.loads(l[-1])['ms_per_step'])")
  echo "prev ms_per_step $ms"
done > $O/ab.txt 2>&1
