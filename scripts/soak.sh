#!/bin/bash
# soak the tests that sit on spin-wait grids, deferred reductions, cross-stream allocator reuse, the data-parallel step and
# graph capture: N repetitions in SHUFFLED order inside one process (LAMP_SOAK, tests/conftest.py), then M fresh processes.  Failing test ids
# and the pass / fail counts go to gpurun_out/soak/soak.txt.   usage: scripts/soak.sh [N=200] [M=20]
set -u
N=${1:-200}; M=${2:-20}
R=$PWD; O=$R/gpurun_out/soak; mkdir -p $O
# (the B = 2048 bf16 ResNet parity test spends minutes in its CPU oracle per repetition: it runs in the fresh-process part only)
SEL=${SOAK_SEL:-'one_pass or bn_backward or batch_norm or deferred or data_parallel or graph or allocator or train_step or stream_and_resume or bn_pair or handoff or statistics or igemm'}
SEL_FRESH="bf16_resnet_step or $SEL"
echo "soak: $N shuffled repetitions in one process, then $M fresh processes; selection: $SEL" > $O/soak.txt
for seed in 1 2; do
  LAMP_SOAK=$((N / 2)) LAMP_SOAK_SEED=$seed timeout ${SOAK_TIMEOUT:-3000} python -m pytest tests -q -m gpu -k "$SEL" -p no:cacheprovider --no-header -rf 2>&1 | grep -E "passed|failed|error|^FAILED|^ERROR" | tail -6 | sed "s/^/in-process seed $seed: /" >> $O/soak.txt
done
pass=0; fail=0
for i in $(seq 1 $M); do
  out=$(timeout 900 python -m pytest tests -q -m gpu -k "$SEL_FRESH" -p no:cacheprovider --no-header -rf 2>&1 | grep -E "passed|failed|error|^FAILED|^ERROR" | tail -4)
  if echo "$out" | grep -qE "failed|error"; then fail=$((fail + 1)); echo "fresh process $i: $out" >> $O/soak.txt; else pass=$((pass + 1)); fi
done
echo "fresh processes: $pass clean, $fail with failures (last summary: ${out:-none})" >> $O/soak.txt
cat $O/soak.txt
