#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc -D flags...>   -> variants/lib_<name>.so
# (conv_f32.hip and conv_halo.hip rebuilt with the flags; every other object of the regular build linked unchanged)
set -e
R=/root/repo; C=$R/casapose_amd/csrc; name=$1; shift
mkdir -p $R/variants
make -C $C -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -munsafe-fp-atomics "$@" -c $C/conv_f32.hip -o /tmp/conv_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -munsafe-fp-atomics "$@" -c $C/conv_halo.hip -o /tmp/halo_$name.o
others=$(for n in capi aux_kernels ls_vote ccl ransac_vote train_kernels conv_wgrad loss_kernels wino wino_gemm guided_bilinear; do echo $C/build/$n.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/conv_$name.o /tmp/halo_$name.o $others -o $R/variants/lib_$name.so
echo built variants/lib_$name.so
