python -m pytest tests/test_ops_gpu.py -m gpu -q -k "igemm or convolution" 2>&1 | tail -5
LAMP_IG_PIPE=2 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "igemm_eight_image" 2>&1 | tail -3
for v in 1 0 2 1 0 2; do LAMP_IG_PIPE=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']
print('PIPE=$v', round(l['ms_per_step'],4), 'ms/step  igemm avg_us', round(r['avg_us'],2), 'frac', round(r['frac'],4))"; done
