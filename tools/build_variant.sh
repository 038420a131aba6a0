#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc -D flags...>   -> variants/lib_<name>.so
set -e
R=/root/repo; C=$R/casapose_amd/csrc; name=$1; shift
mkdir -p $R/variants
make -C $C -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -munsafe-fp-atomics "$@" -c $C/conv_f32.hip -o /tmp/conv_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -munsafe-fp-atomics "$@" -c $C/conv_halo.hip -o /tmp/halo_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/conv_$name.o /tmp/halo_$name.o $C/build/capi.o $C/build/aux_kernels.o $C/build/ls_vote.o $C/build/ccl.o $C/build/ransac_vote.o -o $R/variants/lib_$name.so
echo built variants/lib_$name.so
