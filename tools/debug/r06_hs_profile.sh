#!/bin/bash
# round 6, call 1: where a conv_hsplit tile's time goes on the round-5 tree (HS_PROFILE variant) + the per-layer baseline of the same box
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_hsprof; mkdir -p $O
FILES=conv_hsplit bash tools/build_variant.sh HS_PROFILE -DHS_PROFILE > $O/build.log 2>&1
python tools/layer_times.py --reps 10 > $O/layer_times_base.txt 2>&1
CASAPOSE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_HS_PROFILE.so python tools/debug/hs_profile.py pv_block_3_conv2d pv_block_4_conv2d pv_block_5_conv2d pv_block_6_prepare_conv2d pv_block_7_prepare_conv2d pv_block_8_prepare_conv2d pv_block_9_prepare_conv2d pv_block_10_prepare_conv2d stage1_unit1_conv1 stage1_unit1_conv2 stage2_unit2_conv1 > $O/hs_profile.txt 2>&1
tail -30 $O/hs_profile.txt
