#!/bin/bash
# A/B/C... of several builds of the library on ONE box: bash scripts/ab_libs.sh ROUNDS LIB1 LIB2 [LIB3 ...] -> alternating headline steps (ms)
# extra bench arguments through AB_ARGS (e.g. AB_ARGS="--batch 256")
set -u
P=$1; shift
for i in $(seq 1 $P); do for x in "$@"; do
  ms=$(LAMP_BENCH_ALSO=0 LAMP_LIB_PATH=$x python bench.py --no-cpu-baseline ${AB_ARGS:-} 2>/dev/null | python3 -c "import json,sys; l=[x for x in sys.stdin if x.startswith('{')]; print(json.loads(l[-1])['ms_per_step'])")
  echo "$x ${AB_ARGS:-} ms_per_step $ms"
done; done
