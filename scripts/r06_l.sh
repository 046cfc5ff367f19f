set -u
O=gpurun_out/r06l; mkdir -p $O
hipcc -O3 -std=c++20 --offload-arch=gfx950 scripts/microbench/grid_barrier.hip -o /tmp/gb.bin && timeout 120 /tmp/gb.bin 2>&1 | tee $O/grid_barrier.txt
