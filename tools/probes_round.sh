#!/bin/bash
# ONE GPU-box call for the ISOLATED-kernel probes DESIGN.md argues from (round-4 verdict, weak #7: their outputs used to be one-off files that the
# summary generator deleted).  Every output starts with a stamp line (python3 bench.py --stamp) and lands in gpurun_out/<tag>_probes/; copy them with
#   cp gpurun_out/<tag>_probes/*.txt profiles/probes/        (tools/make_summary.py never touches profiles/probes/)
#   gpurun --timeout 1500 -- 'bash tools/probes_round.sh r05'
# Pieces: pmc_hsplit_isolated.txt  SQ counters of conv_hsplit on four layer shapes, exact split (tile 100) and f16x2 (tile 102)
#         pmc_gemm_isolated.txt    SQ counters of the Winograd-plane GEMM at K = N = 512 / 256 / 128, f16x2 and exact split
#         gemm_power_probe.txt     the K = N = 512 GEMM on random / zero / one operands (same instruction stream, different switching activity)
set -u
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT
tag=${1:-r05}
O=$R/gpurun_out/${tag}_probes
rm -rf $O && mkdir -p $O
STAMP="# stamp $(python3 $R/bench.py --stamp)   ($(date -u +%Y-%m-%dT%H:%MZ), tools/probes_round.sh $tag)"
cd $R
{
  echo "$STAMP"
  echo "# conv_hsplit alone, back to back (tools/pmc_one_layer.sh -> tools/pmc_parse.py): MFMA-busy cycles, clock, instruction mix"
  for mode in 100 102; do
    for spec in "b4 240 320 128 32" "b5 480 640 32 32" "b6 60 80 512 64" "s1 120 160 64 64"; do
      set -- $spec
      bash tools/pmc_one_layer.sh ${tag}_hs_$1_$mode --batch 16 --h $2 --w $3 --cin $4 --cout $5 --dil 1 --tile $mode > /dev/null 2>&1
      echo "== $1: $4 -> $5 at $2 x $3, tile $mode ($([ $mode = 100 ] && echo 'exact bf16 split' || echo 'f16x2'))"
      python3 tools/pmc_parse.py gpurun_out/pmc_${tag}_hs_$1_$mode conv_hsplit 2>&1
      rm -rf gpurun_out/pmc_${tag}_hs_$1_$mode
    done
  done
} > $O/pmc_hsplit_isolated.txt 2>&1
{
  echo "$STAMP"
  for which in f16x2 split; do
    echo "# wino_gemm alone (tools/debug/pmc_gemm.sh), CASAPOSE_GEMM_ONE=$which"
    CASAPOSE_GEMM_ONE=$([ $which = f16x2 ] && echo f16x2 || echo "") bash tools/debug/pmc_gemm.sh 2>&1
    rm -rf gpurun_out/pmc_gemm
  done
} > $O/pmc_gemm_isolated.txt 2>&1
{
  echo "$STAMP"
  echo "# tools/debug/gemm_power_probe.py: the K = N = 512 Winograd-plane GEMM (36 x 5120 rows), 20 launches back to back; identical instruction"
  echo "# streams and memory traffic per row, only the operands' values differ"
  python3 tools/debug/gemm_power_probe.py 2>&1 | grep -v amdgpu.ids
} > $O/gemm_power_probe.txt 2>&1
ls -la $O; head -30 $O/pmc_hsplit_isolated.txt
