O=gpurun_out/r05al; mkdir -p $O
export LAMP_BENCH_ALSO=0
bash scripts/ab_env.sh LAMP_WGRAD_NARROW_PER_CU 1 2 2 > $O/ab_npc2.txt 2>&1
bash scripts/ab_env.sh LAMP_WGRAD_NARROW_PER_CU 1 4 2 > $O/ab_npc4.txt 2>&1
bash scripts/ab_env.sh LAMP_NCV_PER_CU 4 3 2 > $O/ab_ncv3.txt 2>&1
