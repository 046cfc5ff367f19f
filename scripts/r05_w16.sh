#!/bin/bash
# parity of the conv paths with the new kernel, the same-box A/B (previous build vs this one), then the phase clocks of the diagnostic build
export LAMP_BENCH_ALSO=0
mkdir -p gpurun_out/w16
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_resnet_bf16_gpu.py -m gpu -x -q -k "weight or resnet" 2>&1 | tail -3 > gpurun_out/w16/tests.txt
bash scripts/ab_lib.sh ${1:-lamp_amd/lib_prev/liblamp_hip.so} lamp_amd/lib/liblamp_hip.so 4 > gpurun_out/w16/ab.txt 2>&1
LAMP_LIB_PATH=$PWD/lamp_amd/lib_dbg/liblamp_hip.so python scripts/wg8h_stamps.py 2>&1 | tail -3 > gpurun_out/w16/stamps.txt
cat gpurun_out/w16/tests.txt gpurun_out/w16/ab.txt gpurun_out/w16/stamps.txt
