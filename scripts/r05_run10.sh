set -u
O=gpurun_out/r05k; mkdir -p $O
for v in 8 4 2 8 4 2; do
  ms=$(LAMP_BENCH_ALSO=0 LAMP_WGRAD_MIN_IPS=$v python bench.py --no-cpu-baseline --batch 256 2>/dev/null | python3 -c "import json,sys; l=[x for x in sys.stdin if x.startswith('{')]; print(json.loads(l[-1])['ms_per_step'])")
  echo "MIN_IPS=$v b256 $ms"
done 2>&1 | tee $O/ab_ips.txt
cd /tmp; rm -rf /tmp/ks_b256
rocprofv3 --kernel-trace --stats -d /tmp/ks_b256 -o k --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --min-window-s 0 --batch 256 > /tmp/ks_b256.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/trace_step.py $(find /tmp/ks_b256 -name "*kernel_trace.csv" | head -1) > $GRAFT_REPO_ROOT/$O/b256_last_step.txt 2>&1
head -40 $GRAFT_REPO_ROOT/$O/b256_last_step.txt
