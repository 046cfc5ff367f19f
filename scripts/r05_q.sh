O=gpurun_out/r05q; mkdir -p $O
export LAMP_BENCH_ALSO=0
bash scripts/ab_env.sh LAMP_BN_FUSED_NP_MASK 24 31 2 --batch 256 > $O/ab_np_256.txt 2>&1
bash scripts/ab_env.sh LAMP_BN_FUSED_NP_MASK 24 31 2 --batch 32 > $O/ab_np_32.txt 2>&1
bash scripts/ab_env.sh LAMP_BN_FUSED_NP_MASK 24 28 2 > $O/ab_np_2048.txt 2>&1
