set -u
O=gpurun_out/r06g; mkdir -p $O
for pass in 1 2; do for v in lib lib_var/spread lib_var/rot1 lib_var/rot2 lib_var/spreadrot; do
  echo "== $v"; LAMP_LIB_PATH=lamp_amd/$v/liblamp_hip.so python scripts/gemm_ab.py 4096 2>&1 | tail -3
done; done | tee $O/gemm_ab.txt
