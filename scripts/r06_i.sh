set -u
O=gpurun_out/r06i; mkdir -p $O
bash scripts/ab_env.sh 3 LAMP_NCV_PER_CU=4 LAMP_NCV_PER_CU=8 LAMP_NCV_PER_CU=12 2>&1 | tee $O/ab_ncv_per_cu.txt
AB_ARGS="--batch 256" bash scripts/ab_env.sh 2 LAMP_NCV_PER_CU=4 LAMP_NCV_PER_CU=8 2>&1 | tee $O/ab_ncv_per_cu_256.txt
