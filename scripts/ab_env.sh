#!/bin/bash
# A/B of environment settings on ONE box with one library: bash scripts/ab_env.sh ROUNDS "VAR=a" "VAR=b" ... -> alternating headline steps (ms); AB_ARGS: extra bench arguments
set -u
P=$1; shift
for i in $(seq 1 $P); do for x in "$@"; do
  ms=$(env LAMP_BENCH_ALSO=0 $x python bench.py --no-cpu-baseline ${AB_ARGS:-} 2>/dev/null | python3 -c "import json,sys; l=[x for x in sys.stdin if x.startswith('{')]; print(json.loads(l[-1])['ms_per_step'])")
  echo "$x ${AB_ARGS:-} ms_per_step $ms"
done; done
