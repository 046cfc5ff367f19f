#!/bin/bash
# every case of scripts/graph_bisect.py bare and under `rocprofv3 --kernel-trace`; exit codes -> gpurun_out/graphbisect/summary.txt
set -u
ulimit -c 0
R=$PWD; O=$R/gpurun_out/graphbisect; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
: > $O/summary.txt
for c in ${CASES:-ew3 ew200 zeros sum conv_small_bf16 conv_bf16 conv_f32 conv_bwd_bf16 conv_bwd_f32 bn_bf16 bn_f32 mlp resnet_f32 resnet_bf16 resnet_bf16_b2048 resnet_bf16_b2048_opt}; do
  timeout 300 python3 $R/scripts/graph_bisect.py $c > $O/bare_$c.log 2>&1; rb=$?
  rm -rf /tmp/gb_$c
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gb_$c -o g -- python3 $R/scripts/graph_bisect.py $c > $O/prof_$c.log 2>&1; rp=$?
  echo "$c: bare rc $rb, under rocprofv3 --kernel-trace rc $rp $(grep -m1 -E 'SIGSEGV|SIGABRT|Aborted' $O/prof_$c.log | cut -c1-80)" >> $O/summary.txt
done
# the bench itself on the same box, graph forced on under the profiler
for i in 1 2 3; do
  rm -rf /tmp/gb_bench
  LAMP_BENCH_GRAPH_UNDER_PROFILER=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gb_bench -o g -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $O/prof_bench_$i.log 2>&1
  echo "bench.py (graph forced on) run $i: under rocprofv3 --kernel-trace rc $? $(grep -m1 -E 'SIGSEGV|SIGABRT|Aborted' $O/prof_bench_$i.log | cut -c1-80)" >> $O/summary.txt
done
cat $O/summary.txt
