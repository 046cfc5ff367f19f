set -u
O=gpurun_out/r05aq; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "conv or statistics" 2>&1 | tail -8 > $O/pytest_a.txt
timeout 900 python -m pytest tests/test_resnet_bf16_gpu.py -q -m gpu -x 2>&1 | tail -5 > $O/pytest_b.txt
export LAMP_BENCH_ALSO=0
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; l=[x for x in sys.stdin if x.startswith('{')]; print(json.loads(l[-1])['ms_per_step'])"; done > $O/ms.txt
