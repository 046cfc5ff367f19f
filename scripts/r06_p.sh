set -u
O=gpurun_out/r06p; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "weight_grad or convolution_forward_backward or narrow or pair" 2>&1 | tail -4 | tee $O/pytest_sel.txt
bash scripts/ab_env.sh 3 "LAMP_NCV_WG_PITCH=0" "LAMP_NCV_WG_PITCH=1" 2>&1 | tee $O/ab.txt
bash scripts/prof_resnet.sh > /dev/null 2>&1; grep "ncv_wgrad2" gpurun_out/prof/timeline.txt | tee $O/ncv_timeline.txt
