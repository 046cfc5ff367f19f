O=gpurun_out/r05z; mkdir -p $O
export LAMP_BENCH_ALSO=0
bash scripts/ab_lib.sh $PWD/lamp_amd/lib/liblamp_hip.so $PWD/lamp_amd/lib_b4/liblamp_hip.so 3 > $O/ab_2048.txt 2>&1
bash scripts/ab_lib.sh $PWD/lamp_amd/lib/liblamp_hip.so $PWD/lamp_amd/lib_b4/liblamp_hip.so 2 --batch 256 > $O/ab_256.txt 2>&1
