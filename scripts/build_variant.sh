#!/bin/bash
# bash scripts/build_variant.sh NAME "-DFLAG=.." -> lamp_amd/lib_prev/liblamp_hip_NAME.so (conv_igemm.hip rebuilt with the flags, other objects from build/)
set -e
cd lamp_amd/csrc; mkdir -p build_var ../lib_prev
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -I../../include --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-result $2 -include kernels/abi_dev_names.h -c kernels/conv_igemm.hip -o build_var/conv_igemm_$1.o
OBJS=$(find build -name '*.o' | grep -v kernels/conv_igemm.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o ../lib_prev/liblamp_hip_$1.so $OBJS build_var/conv_igemm_$1.o
