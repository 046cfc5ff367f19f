#!/bin/bash
# Runs on the GPU box (gpurun): full GPU test suite, bench line, rocprofv3 kernel stats, the two PMC passes and the secondary probes.
# Everything judged is written under gpurun_out/refresh/ and copied into profiles/ by hand afterwards.
set -u
R=$PWD; O=$R/gpurun_out/refresh; mkdir -p $O; RND=${RND:-r03}
export TMPDIR=/tmp
ulimit -c 0
[ -n "${SKIP_PYTEST:-}" ] || timeout 1800 python -m pytest tests -q -m gpu -rf 2>&1 | grep -E "passed|failed|error|^FAILED" | tail -8 > $O/pytest_gpu.txt
cd /tmp
# (rocprofv3 around the graph-replaying bench died with SIGSEGV in 2 of 15 runs on this pool - never without the profiler: retry)
for attempt in 1 2 3; do
  rm -rf /tmp/ks
  rocprofv3 --kernel-trace --stats -d /tmp/ks -o k --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > /tmp/ks.log 2>&1 && break
done
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 $R/scripts/trace_step.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) > $O/last_step_breakdown.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o f -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > /tmp/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o w -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > /tmp/pw.log 2>&1
python3 $R/scripts/pmc_traffic.py $(find /tmp/pf -name "*counter_collection.csv" | head -1) $(find /tmp/pw -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
cp $O/pmc_traffic.json $R/profiles/${RND}_pmc_traffic.json   # the bench line below reads its roofline.traffic from this run's passes
cd $R; python bench.py > $O/bench.log 2>$O/bench.err; cd /tmp
for w in gemm knn attention umap lm; do
  rocprofv3 --kernel-trace --stats -d /tmp/ks_$w -o k --output-format csv -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > /tmp/ks_$w.log 2>&1
  cp $(find /tmp/ks_$w -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$w.csv
  grep -o '{"metric.*' /tmp/ks_$w.log | tail -1 > $O/bench_$w.log
done
cd $R
# the un-profiled lines of the secondary workloads WITH their CPU baselines (kNN / UMAP: bounded samples, oracle/cpu_baseline.py)
for w in knn umap umap-e2e mlp; do python bench.py --workload $w > $O/bench_full_$w.log 2>/dev/null; done
LAMP_LIB_PATH=lamp_amd/lib_stamp/liblamp_hip.so python scripts/gemm_clock_probe.py > $O/gemm_clock.txt 2>/dev/null || true
{ python scripts/attn_probe.py 8 16 4096 128 0; python scripts/attn_probe.py 8 16 4096 128 1; python scripts/attn_probe.py 8 16 4096 64 0; } > $O/attention_probe.txt 2>&1
python scripts/umap_full_probe.py 1000000 40 2>&1 | grep "umap n" > $O/umap_probe.txt
python scripts/gemm_ab.py > $O/gemm_ab.txt 2>&1 || true
python scripts/tf_ops_probe.py > $O/tf_ops_probe.txt 2>&1 || true
python scripts/kstats_top.py /tmp/ks_lm 40 6 > $O/lm_kernels.txt 2>&1 || true
cat $O/pytest_gpu.txt; tail -c 600 $O/bench.log; cat $O/pmc_traffic.txt | head -12; cat $O/umap_probe.txt
