#!/bin/bash
# the graph-replayed training step under `rocprofv3 --kernel-trace` (bench.py normally turns the graph off there: forced on
# with LAMP_BENCH_GRAPH_UNDER_PROFILER=1), R runs per variant; exit codes and signals -> gpurun_out/graphrepro/summary.txt
set -u
ulimit -c 0
R=$PWD; O=$R/gpurun_out/graphrepro; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
RUNS=${1:-8}
variant() {   # name, then NAME=VALUE settings (exported to the program through the environment of this shell: the program itself follows `--`)
  name=$1; shift
  ok=0; bad=0
  for i in $(seq 1 $RUNS); do
    rm -rf /tmp/gr_$name
    ( for kv in "$@"; do export "$kv"; done; export LAMP_BENCH_GRAPH_UNDER_PROFILER=1
      rocprofv3 --kernel-trace --output-format csv -d /tmp/gr_$name -o g -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $O/${name}_$i.log 2>&1 )
    rc=$?
    if [ $rc -eq 0 ] && grep -q '"metric"' $O/${name}_$i.log; then ok=$((ok + 1)); rm -f $O/${name}_$i.log; else bad=$((bad + 1)); echo "$name run $i: rc $rc: $(grep -m1 -iE 'segmentation|signal|fault|abort|error' $O/${name}_$i.log | cut -c1-200)" >> $O/summary.txt; fi
  done
  echo "$name: $ok ok, $bad failed of $RUNS ($*)" >> $O/summary.txt
}
: > $O/summary.txt
variant graph_default
if grep -q "graph_default: .* [1-9][0-9]* failed" $O/summary.txt; then
  variant kernarg_host HIP_FORCE_DEV_KERNARG=0
  variant no_deferred_reduce LAMP_DEFER_WGRAD_REDUCE=0
fi
cat $O/summary.txt
