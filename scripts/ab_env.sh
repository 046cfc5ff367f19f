#!/bin/bash
# A/B of one environment switch on ONE box: bash scripts/ab_env.sh VAR A B [pairs] [bench args...] -> alternating headline steps (ms) per setting
set -u
V=$1; A=$2; B=$3; P=${4:-3}; shift 4 || shift $#
for i in $(seq 1 $P); do for x in "$A" "$B"; do
  ms=$(env $V=$x python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; l=[x for x in sys.stdin if x.startswith('{')]; print(json.loads(l[-1])['ms_per_step'])")
  echo "$V=$x ms_per_step $ms"
done; done
