#!/bin/bash
# Round 5, first look: ordered timeline of the headline step and the small-batch lines (bf16 / f32 at B = 256 and 32) with per-kernel tables.
set -u
R=$PWD; O=$R/gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
python bench.py --no-cpu-baseline > $O/b2048_bench.log 2>$O/b2048_bench.err
for B in 256 32; do for D in bf16 f32; do
  python bench.py --batch $B --dtype $D --no-cpu-baseline > $O/b${B}_${D}_bench.log 2>$O/b${B}_${D}_bench.err
  python bench.py --batch $B --dtype $D --no-cpu-baseline --no-graph > $O/b${B}_${D}_eager_bench.log 2>/dev/null
done; done
cd /tmp
prof() { # name, bench args...
  n=$1; shift
  rm -rf /tmp/ks_$n
  rocprofv3 --kernel-trace --stats -d /tmp/ks_$n -o k --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --min-window-s 0 "$@" > $O/${n}_rocprof.log 2>&1
  cp $(find /tmp/ks_$n -name "*kernel_stats.csv" | head -1) $O/${n}_kernel_stats.csv
  python3 $R/scripts/trace_step.py $(find /tmp/ks_$n -name "*kernel_trace.csv" | head -1) > $O/${n}_last_step.txt 2>&1
  python3 $R/scripts/trace_step.py $(find /tmp/ks_$n -name "*kernel_trace.csv" | head -1) --timeline > $O/${n}_timeline.txt 2>&1
}
prof b2048
prof b256 --batch 256
prof b32 --batch 32
prof b256_f32 --batch 256 --dtype f32
prof b2048_f32 --dtype f32
cd $R
tail -c 300 $O/b2048_bench.log; for f in $O/b*_bench.log; do echo $f; python3 -c "
import json,sys
l=[x for x in open('$f') if x.startswith('{')]
d=json.loads(l[-1]); print(d['ms_per_step'], d['value'])"; done
