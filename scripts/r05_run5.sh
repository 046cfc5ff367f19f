set -u
O=gpurun_out/r05f; mkdir -p $O
timeout 300 python -m pytest tests/test_data.py -q -x -k "pinned" 2>&1 | tail -15 > $O/pytest_pinned.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv" 2>&1 | tail -5 > $O/pytest_conv.txt
timeout 600 python -m pytest tests/test_resnet_bf16_gpu.py -q -x 2>&1 | tail -5 >> $O/pytest_conv.txt
cat $O/pytest_pinned.txt $O/pytest_conv.txt
LAMP_BENCH_ALSO=0 bash scripts/ab_env.sh LAMP_IG_RING4 0 1 2 --batch 256 2>&1 | tee $O/ab_ring4_b256.txt
LAMP_BENCH_ALSO=0 bash scripts/ab_env.sh LAMP_IG_RING4 0 1 2 --batch 32 2>&1 | tee $O/ab_ring4_b32.txt
# counter-free evidence for VERDICT r4 item 6: the reductions right behind their producers, per-launch durations
export LAMP_DEFER_WGRAD_REDUCE=0
bash scripts/prof_resnet.sh > /dev/null 2>&1; cp gpurun_out/prof/timeline.txt $O/timeline_immediate_reduce.txt; cp gpurun_out/prof/last_step.txt $O/last_step_immediate_reduce.txt
unset LAMP_DEFER_WGRAD_REDUCE
grep -E "wgrad" $O/timeline_immediate_reduce.txt | head -40
