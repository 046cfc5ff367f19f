#!/bin/bash
# Round-5 additions to scripts/refresh_profiles.sh (run first with RND=r05 SKIP_PYTEST=1): the small-batch lines VERDICT r4 item 3 asks for
# (B = 256 and 32, bf16 and f32, bench line + rocprofv3 last-step table), the f32 / f64 steps with their kernel tables, the epoch and MLP
# lines, and the SQ counter passes with the repaired summary.  Everything lands under gpurun_out/refresh5/ and is copied into profiles/r05_*.
set -u
R=$PWD; O=$R/gpurun_out/refresh5; mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
for B in 256 32; do for D in bf16 f32; do
  LAMP_BENCH_ALSO=0 python bench.py --batch $B --dtype $D --no-cpu-baseline > $O/resnet_b${B}_${D}_bench.log 2>/dev/null
done; done
python bench.py --dtype f32 > $O/resnet_f32_bench.log 2>$O/resnet_f32_bench.err
python bench.py --dtype f64 --no-cpu-baseline > $O/resnet_f64_bench.log 2>$O/resnet_f64_bench.err
python bench.py --workload epoch > $O/epoch_bench.log 2>$O/epoch_bench.err
python bench.py --workload mlp > $O/mlp_bench.log 2>/dev/null
cd /tmp
prof() { # name, bench args...
  n=$1; shift
  rm -rf /tmp/ks_$n
  rocprofv3 --kernel-trace --stats -d /tmp/ks_$n -o k --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --min-window-s 0 "$@" > /tmp/ks_$n.log 2>&1
  cp $(find /tmp/ks_$n -name "*kernel_stats.csv" | head -1) $O/${n}_kernel_stats.csv
  python3 $R/scripts/trace_step.py $(find /tmp/ks_$n -name "*kernel_trace.csv" | head -1) > $O/${n}_last_step_breakdown.txt 2>&1
}
prof resnet_step_eager
python3 $R/scripts/trace_step.py $(find /tmp/ks_resnet_step_eager -name "*kernel_trace.csv" | head -1) --timeline > $O/resnet_step_timeline.txt 2>&1
prof resnet_b256 --batch 256
prof resnet_b32 --batch 32
prof resnet_b256_f32 --batch 256 --dtype f32
prof resnet_f32 --dtype f32
prof resnet_f64 --dtype f64
cd $R
bash scripts/sq_counters.sh > $O/sq_run.log 2>&1
cp gpurun_out/sq/summary.txt $O/sq_summary.txt 2>/dev/null
for f in gpurun_out/sq/*_group*.txt; do cp $f $O/sq_$(basename $f); done
grep -i -E "mall|TCC_EA_RD|TCC_HIT|TCC_MISS" gpurun_out/sq/counters_available.txt | head -40 > $O/counters_cache_related.txt 2>/dev/null
for f in $O/resnet_b*_bench.log $O/resnet_f32_bench.log $O/resnet_f64_bench.log; do echo $f; python3 -c "
import json
l=[x for x in open('$f') if x.startswith('{')]
d=json.loads(l[-1]); print(d['ms_per_step'], d['value'])"; done
head -20 $O/sq_summary.txt
