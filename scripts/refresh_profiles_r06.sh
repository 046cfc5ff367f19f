#!/bin/bash
# Round 6: everything judged, on ONE box - the full GPU test suite, the headline line, rocprofv3 kernel statistics of the (eager) step with the
# per-class table the bench line quotes, the two PMC passes, the small-batch / f32 / f64 tables, and the secondary workloads.
# Output under gpurun_out/refresh6/; scripts/collect_profiles_r06.sh copies what is judged into profiles/r06_*.
set -u
R=$PWD; O=$R/gpurun_out/refresh6; mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
[ -n "${SKIP_PYTEST:-}" ] || timeout 2400 python -m pytest tests -q -m gpu -rf 2>&1 | grep -E "passed|failed|error|^FAILED|^ERROR" | tail -12 > $O/pytest_gpu.txt
cd /tmp
prof() { # name, bench args...
  n=$1; shift
  rm -rf /tmp/ks_$n
  rocprofv3 --kernel-trace --stats -d /tmp/ks_$n -o k --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --min-window-s 0 "$@" > /tmp/ks_$n.log 2>&1
  cp $(find /tmp/ks_$n -name "*kernel_stats.csv" | head -1) $O/${n}_kernel_stats.csv
  python3 $R/scripts/trace_step.py $(find /tmp/ks_$n -name "*kernel_trace.csv" | head -1) > $O/${n}_last_step_breakdown.txt 2>&1
}
prof resnet_step
python3 $R/scripts/trace_step.py $(find /tmp/ks_resnet_step -name "*kernel_trace.csv" | head -1) --timeline --geometry > $O/resnet_step_timeline.txt 2>&1
python3 $R/scripts/class_rocprof.py $O/resnet_step_kernel_stats.csv $O/class_rocprof.json "rocprofv3 --kernel-trace --stats of bench.py --no-graph --steps 10 (the eager step: rocprofv3 and hipGraphLaunch cannot be combined on this image; the timed region replays the graph)" > $O/class_rocprof.txt 2>&1
cp $O/class_rocprof.json $R/profiles/r06_class_rocprof.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o f -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > /tmp/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o w -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > /tmp/pw.log 2>&1
python3 $R/scripts/pmc_traffic.py $(find /tmp/pf -name "*counter_collection.csv" | head -1) $(find /tmp/pw -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
cp $O/pmc_traffic.json $R/profiles/r06_pmc_traffic.json
cd $R; python bench.py > $O/bench.log 2>$O/bench.err; cd /tmp
prof resnet_b256 --batch 256
prof resnet_b32 --batch 32
prof resnet_f32 --dtype f32
for w in gemm knn attention umap lm; do
  rocprofv3 --kernel-trace --stats -d /tmp/ks_$w -o k --output-format csv -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > /tmp/ks_$w.log 2>&1
  cp $(find /tmp/ks_$w -name "*kernel_stats.csv" | head -1) $O/${w}_kernel_stats.csv
done
cd $R
for w in gemm knn attention umap umap-e2e lm mlp epoch; do python bench.py --workload $w > $O/${w}_bench.log 2>/dev/null; done
python scripts/gemm_ab.py > $O/gemm_ab.txt 2>&1 || true
cat $O/pytest_gpu.txt; tail -c 700 $O/bench.log; head -12 $O/pmc_traffic.txt; cat $O/class_rocprof.txt
