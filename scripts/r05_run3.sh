set -u
O=gpurun_out/r05d; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "two_workgroups_of_four or convolution_pair" 2>&1 | tail -8 > $O/pytest_a.txt
timeout 300 python -m pytest tests/test_data.py -q -x -k "pinned" 2>&1 | tail -4 >> $O/pytest_a.txt
cat $O/pytest_a.txt
for d in 0 4000 9000 14000 20000; do
  for m in 0 1; do
    ms=$(LAMP_IG_4E=$m LAMP_IG_4E_DELAY=$d LAMP_BENCH_ALSO=0 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; l=[x for x in sys.stdin if x.startswith('{')]; print(json.loads(l[-1])['ms_per_step'])")
    echo "IG_4E=$m delay=$d ms_per_step $ms"
  done
done 2>&1 | tee $O/ab_4e.txt
