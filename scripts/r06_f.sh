set -u
O=gpurun_out/r06f; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 scripts/microbench/gemm_kstep_loop.hip -o /tmp/gk.bin && /tmp/gk.bin 2>&1 | tee $O/gemm_kstep_loop.txt
