set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
for pb in 256 224 192 160 128; do CASAPOSE_PERSIST_BLOCKS=$pb timeout 300 python tools/debug/overlap_probe.py >> $O/overlap.txt 2>&1; done
grep persist $O/overlap.txt
