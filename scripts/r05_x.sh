set -u
O=gpurun_out/r05x; mkdir -p $O
R=$PWD; export TMPDIR=/tmp LAMP_BENCH_ALSO=0
cd /tmp
for B in 2048; do
rm -rf /tmp/ks_$B
rocprofv3 --kernel-trace --stats -d /tmp/ks_$B -o k --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --min-window-s 0 --batch $B > /tmp/ks_$B.log 2>&1
python3 $R/scripts/trace_step.py $(find /tmp/ks_$B -name "*kernel_trace.csv" | head -1) --timeline > $R/$O/timeline_$B.txt 2>&1
done
