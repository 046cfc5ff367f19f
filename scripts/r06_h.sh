set -u
O=gpurun_out/r06h; mkdir -p $O
LAMP_LIB_PATH=lamp_amd/lib_var/ncvstamp/liblamp_hip.so python scripts/ncv_stamp_probe.py 2>&1 | tee $O/ncv_stamps.txt
bash scripts/prof_resnet.sh > $O/prof_head.txt 2>&1; cp gpurun_out/prof/timeline.txt $O/timeline.txt; cp gpurun_out/prof/kernel_stats.csv $O/kernel_stats.csv
