#!/bin/bash
# usage: [FILES="conv_f32 conv_halo"] tools/build_variant.sh <name> <extra hipcc -D flags...>   -> variants/lib_<name>.so
# (the listed sources -- default conv_f32.hip and conv_halo.hip -- rebuilt with the flags; every other object of the regular build linked unchanged)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; C=$R/casapose_amd/csrc; name=$1; shift
FILES=${FILES:-"conv_f32 conv_halo"}
mkdir -p $R/variants
make -C $C -s
objs=""
for n in capi conv_f32 conv_halo conv_stem aux_kernels ls_vote ccl ransac_vote train_kernels conv_wgrad loss_kernels wino wino_gemm wino_gemm_split wino_gemm_wide guided_bilinear conv_hsplit conv_wgrad_split head1x1 loss_functional wino_wgrad_split conv_bf16d conv_stem_split; do
  if echo " $FILES " | grep -q " $n "; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -munsafe-fp-atomics -fno-slp-vectorize "$@" -c $C/$n.hip -o /tmp/${n}_$name.o
    objs="$objs /tmp/${n}_$name.o"
  else
    objs="$objs $C/build/$n.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/variants/lib_$name.so
echo built variants/lib_$name.so
