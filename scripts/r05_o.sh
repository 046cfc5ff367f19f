set -u
O=gpurun_out/r05o; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "statistics_from or convolution_pair" 2>&1 | tail -4 > $O/pytest_a.txt
timeout 600 python -m pytest tests/test_resnet_bf16_gpu.py -q -m gpu -x 2>&1 | tail -3 > $O/pytest_b.txt
export LAMP_BENCH_ALSO=0
bash scripts/ab_env.sh LAMP_NCV_BN_STATS 0 1 3 > $O/ab_2048.txt 2>&1
bash scripts/ab_env.sh LAMP_NCV_BN_STATS 0 1 2 --batch 256 > $O/ab_256.txt 2>&1
R=$PWD; export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ks_1
rocprofv3 --kernel-trace --stats -d /tmp/ks_1 -o k --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --min-window-s 0 > /tmp/ks_1.log 2>&1
python3 $R/scripts/trace_step.py $(find /tmp/ks_1 -name "*kernel_trace.csv" | head -1) --timeline > $R/$O/timeline_1.txt 2>&1
