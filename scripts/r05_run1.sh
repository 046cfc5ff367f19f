set -u
mkdir -p gpurun_out/r05b
python -m pytest tests/test_ops_gpu.py -q -x -k "convolution_pair or statistics_from_the_convolution or convolution" 2>&1 | tail -15 > gpurun_out/r05b/pytest_pair.txt
python -m pytest tests/test_resnet_bf16_gpu.py tests/test_autograd_gpu.py -q -x 2>&1 | tail -15 > gpurun_out/r05b/pytest_resnet.txt
bash scripts/ab_env.sh LAMP_CONV_SIBLING 0 1 3 > gpurun_out/r05b/ab_sibling.txt 2>&1
bash scripts/ab_env.sh LAMP_CONV_SIBLING 0 1 2 --batch 256 > gpurun_out/r05b/ab_sibling_b256.txt 2>&1
cat gpurun_out/r05b/pytest_pair.txt gpurun_out/r05b/pytest_resnet.txt gpurun_out/r05b/ab_sibling.txt gpurun_out/r05b/ab_sibling_b256.txt
