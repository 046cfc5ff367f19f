#!/bin/bash
# rocprofv3 kernel trace of the ResNet bench step (eager: replayed graphs hide kernel names) -> gpurun_out/prof/{kernel_stats.csv,last_step.txt,timeline.txt}
set -u
R=$PWD; O=$R/gpurun_out/prof; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats -d /tmp/ks -o k --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --min-window-s 0 "$@" > $O/bench_under_rocprof.log 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 $R/scripts/trace_step.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) > $O/last_step.txt 2>&1
python3 $R/scripts/trace_step.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) --timeline --geometry > $O/timeline.txt 2>&1
head -45 $O/last_step.txt
