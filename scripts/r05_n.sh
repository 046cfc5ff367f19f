set -u
R=$PWD; O=$R/gpurun_out/r05n; mkdir -p $O
export TMPDIR=/tmp LAMP_BENCH_ALSO=0
cd /tmp
for v in 0 1; do
  export LAMP_NCV_BN_STATS=$v
  rm -rf /tmp/ks_$v
  rocprofv3 --kernel-trace --stats -d /tmp/ks_$v -o k --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --min-window-s 0 > /tmp/ks_$v.log 2>&1
  python3 $R/scripts/trace_step.py $(find /tmp/ks_$v -name "*kernel_trace.csv" | head -1) --timeline > $O/timeline_$v.txt 2>&1
done
