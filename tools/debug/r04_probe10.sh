set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
for bl in 256 224 192; do for sk in 0.0 0.3 0.5; do CASAPOSE_TWO_STREAM_BLOCKS=$bl CASAPOSE_TWO_STREAM_SKEW=$sk timeout 300 python tools/debug/two_stream_bench.py >> $O/two.txt 2>&1; done; done
grep -h "blocks\|bit-equal" $O/two.txt | sort | uniq -c | sort -rn | head -30
