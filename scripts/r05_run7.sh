set -u
O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "mode_unique or accumulates_in_the_kernel" 2>&1 | tail -12 | tee $O/pytest_a.txt
timeout 900 python -m pytest tests/test_autograd_gpu.py -q -x -k "resnet" 2>&1 | tail -4 | tee -a $O/pytest_a.txt
for d in f32 f64; do LAMP_BENCH_ALSO=0 python bench.py --no-cpu-baseline --dtype $d 2>/dev/null | python3 -c "import json,sys; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print('$d', d['ms_per_step'])"; done | tee $O/fp_steps.txt
cd /tmp; rm -rf /tmp/ks32
rocprofv3 --kernel-trace --stats -d /tmp/ks32 -o k --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --dtype f32 --steps 5 --warmup 3 --no-cpu-baseline --no-graph --min-window-s 0 > /tmp/ks32.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/trace_step.py $(find /tmp/ks32 -name "*kernel_trace.csv" | head -1) > $GRAFT_REPO_ROOT/$O/f32_last_step.txt 2>&1
head -30 $GRAFT_REPO_ROOT/$O/f32_last_step.txt
