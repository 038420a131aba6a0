set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q; mkdir -p $O
./variants/f16_probe > $O/f16_probe.txt 2>&1
python tools/debug/gemm_planes_probe.py > $O/planes_bf16.txt 2>&1
CASAPOSE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_f16timing.so python tools/debug/gemm_planes_probe.py > $O/planes_f16.txt 2>&1
cat $O/f16_probe.txt; grep -h "sum\|512->512\|128->128" $O/planes_bf16.txt; echo f16; grep -h "sum\|512->512\|128->128" $O/planes_f16.txt
