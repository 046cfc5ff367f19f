#!/bin/bash
# VERDICT r3 item 2(b), third stage: is it the NUMBER of graph launches / kernel records?  Every line: program, replays, exit code under
# `rocprofv3 --kernel-trace`.  -> gpurun_out/graphvolume/summary.txt
set -u
ulimit -c 0      # a faulting run must not spend minutes writing a core file of a process with the GPU mapped
R=$PWD; O=$R/gpurun_out/graphvolume; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
[ "${GV_PART:-1}" = 1 ] && : > $O/summary.txt
run() {  # label, then the program and its arguments
  label=$1; shift
  rm -rf /tmp/gv
  timeout -s KILL ${GV_TIMEOUT:-150} rocprofv3 --kernel-trace --output-format csv -d /tmp/gv -o g -- "$@" > $O/$label.log 2>&1; rc=$?
  n=$(cat $(find /tmp/gv -name "*kernel_trace.csv" 2>/dev/null | head -1) 2>/dev/null | wc -l)
  echo "$label: rc $rc, kernel records written $n $(grep -m1 -E 'SIGSEGV' $O/$label.log | cut -c1-60)" >> $O/summary.txt
}
export LAMP_BENCH_GRAPH_UNDER_PROFILER=1
if [ "${GV_PART:-1}" = 1 ]; then
run bench_8_launches python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --min-window-s 0
run bench_40_launches python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --min-window-s 0
run bench_150_launches python3 $R/bench.py --steps 140 --warmup 10 --no-cpu-baseline --min-window-s 0
run bench_default_window python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline
run toy_3_kernels_x_2000 python3 $R/scripts/graph_toy.py 2000
run toy_3_kernels_x_30000 python3 $R/scripts/graph_toy.py 30000
run ew200_x_50 python3 $R/scripts/graph_bisect.py ew200 50
run ew200_x_500 python3 $R/scripts/graph_bisect.py ew200 500
run resnet_b2048_x_50 python3 $R/scripts/graph_bisect.py resnet_bf16_b2048_opt 50
run resnet_b2048_x_300 python3 $R/scripts/graph_bisect.py resnet_bf16_b2048_opt 300
run resnet_b2048_x_1000 python3 $R/scripts/graph_bisect.py resnet_bf16_b2048_opt 1000
fi
if [ "${GV_PART:-1}" = 2 ]; then
run toy_3_kernels_x_30000 python3 $R/scripts/graph_toy.py 30000
run ew200_x_100 python3 $R/scripts/graph_bisect.py ew200 100
run ew200_x_500 python3 $R/scripts/graph_bisect.py ew200 500
run resnet_b2048_x_150 python3 $R/scripts/graph_bisect.py resnet_bf16_b2048_opt 150
run resnet_b2048_x_600 python3 $R/scripts/graph_bisect.py resnet_bf16_b2048_opt 600
run bench_300_launches python3 $R/bench.py --steps 290 --warmup 10 --no-cpu-baseline --min-window-s 0
run bench_600_launches python3 $R/bench.py --steps 590 --warmup 10 --no-cpu-baseline --min-window-s 0
fi
cat $O/summary.txt
