set -u
O=gpurun_out/r05c; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -q -x -k "convolution_pair or sorting_family or randperm or unique or sort_argsort" 2>&1 | tail -8 > $O/pytest_a.txt
python -m pytest tests/test_resnet_bf16_gpu.py tests/test_data.py -q -x 2>&1 | tail -8 > $O/pytest_b.txt
bash scripts/ab_lib.sh lamp_amd/lib_a/liblamp_hip.so lamp_amd/lib/liblamp_hip.so 3 > $O/ab_v1_v2.txt 2>&1
bash scripts/ab_env.sh LAMP_CONV_SIBLING 0 1 2 > $O/ab_sibling.txt 2>&1
bash scripts/prof_resnet.sh > /dev/null 2>&1; cp gpurun_out/prof/last_step.txt $O/last_step.txt; cp gpurun_out/prof/timeline.txt $O/timeline.txt
cat $O/pytest_a.txt $O/pytest_b.txt $O/ab_v1_v2.txt $O/ab_sibling.txt; head -30 $O/last_step.txt
