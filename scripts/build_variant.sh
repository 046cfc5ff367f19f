#!/bin/bash
# bash scripts/build_variant.sh NAME "-DFLAG=.." [kernels/FILE.hip] -> lamp_amd/lib_var/NAME/liblamp_hip.so (that source rebuilt with the flags, other objects from build/)
set -e
SRC=${3:-kernels/conv_igemm.hip}
B=$(basename $SRC .hip)
cd lamp_amd/csrc; mkdir -p build_var ../lib_var/$1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -I../../include --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-result $2 -include kernels/abi_dev_names.h -c $SRC -o build_var/${B}_$1.o
OBJS=$(find build -name '*.o' | grep -v kernels/$B.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o ../lib_var/$1/liblamp_hip.so $OBJS build_var/${B}_$1.o
