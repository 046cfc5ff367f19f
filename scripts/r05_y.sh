O=gpurun_out/r05y; mkdir -p $O
export LAMP_BENCH_ALSO=0
bash scripts/ab_env.sh LAMP_WGRAD_MIN_IPS 2 4 3 --batch 256 > $O/ab_ips_256.txt 2>&1
bash scripts/ab_env.sh LAMP_WGRAD_MIN_IPS 2 4 2 --batch 32 > $O/ab_ips_32.txt 2>&1
bash scripts/ab_env.sh LAMP_WGRAD_MIN_IPS 2 4 2 --batch 512 > $O/ab_ips_512.txt 2>&1
