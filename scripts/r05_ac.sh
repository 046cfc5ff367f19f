O=gpurun_out/r05ac; mkdir -p $O
export LAMP_BENCH_ALSO=0
bash scripts/ab_env.sh LAMP_BN_FUSED_SMALL_BYTES 4194304 4194305 3 --batch 256 > $O/ab_256.txt 2>&1
bash scripts/ab_env.sh LAMP_BN_FUSED_SMALL_BYTES 4194304 4194305 3 > $O/ab_2048.txt 2>&1
bash scripts/ab_env.sh LAMP_BN_FUSED_SMALL_BYTES 4194304 8388609 2 > $O/ab_2048b.txt 2>&1
