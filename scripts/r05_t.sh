set -u
O=gpurun_out/r05t; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "two_first_convolutions or input_gradient" 2>&1 | tail -12 > $O/pytest_a.txt
R=$PWD; export TMPDIR=/tmp LAMP_BENCH_ALSO=0
cd /tmp
rm -rf /tmp/ks_1
rocprofv3 --kernel-trace --stats -d /tmp/ks_1 -o k --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --min-window-s 0 > /tmp/ks_1.log 2>&1
python3 $R/scripts/trace_step.py $(find /tmp/ks_1 -name "*kernel_trace.csv" | head -1) --timeline > $R/$O/timeline.txt 2>&1
