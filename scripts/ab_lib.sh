#!/bin/bash
# A/B of two builds of the library on ONE box: bash scripts/ab_lib.sh LIB_A LIB_B [pairs] [bench args...] -> alternating headline steps (ms)
set -u
A=$1; B=$2; P=${3:-3}; shift 3 || shift $#
for i in $(seq 1 $P); do for x in "$A" "$B"; do
  ms=$(LAMP_LIB_PATH=$x python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; l=[x for x in sys.stdin if x.startswith('{')]; print(json.loads(l[-1])['ms_per_step'])")
  echo "$x ms_per_step $ms"
done; done
