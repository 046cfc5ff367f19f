set -u
O=gpurun_out/r06k; mkdir -p $O
bash scripts/ab_env.sh 3 LAMP_BN_FUSED_NP_MASK=24 LAMP_BN_FUSED_NP_MASK=28 LAMP_BN_FUSED_NP_MASK=30 2>&1 | tee $O/ab_np_mask.txt
