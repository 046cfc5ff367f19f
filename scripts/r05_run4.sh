set -u
O=gpurun_out/r05e; mkdir -p $O
timeout 300 python -m pytest tests/test_data.py -q -x -k "pinned" 2>&1 | tail -40 > $O/pytest_pinned.txt
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "convolution_pair or deferred or sorting_family" 2>&1 | tail -5 > $O/pytest_a.txt
LAMP_BENCH_ALSO=0 bash scripts/ab_env.sh LAMP_DEFER_WGRAD_REDUCE 1 0 3 > $O/ab_defer.txt 2>&1
python bench.py --no-cpu-baseline > $O/bench_also.log 2>$O/bench_also.err
cat $O/pytest_pinned.txt $O/pytest_a.txt $O/ab_defer.txt; python3 -c "
import json
l=[x for x in open('$O/bench_also.log') if x.startswith('{')]; d=json.loads(l[-1]); print(d['ms_per_step'], d.get('also'))"
