#!/bin/bash
# ONE GPU-box call for the round-6 probe outputs DESIGN.md argues from -> gpurun_out/r06_probes/ -> copy into profiles/probes/ (each file starts with a stamp line)
#   hsplit_trace.txt     time line of conv_hsplit's two roles per barrier phase (HS_TRACE variant; build it here first:
#                        FILES=conv_hsplit bash tools/build_variant.sh HS_TRACE -DHS_TRACE)
#   wino_parts.txt       input transform / GEMM / output transform of every Winograd layer, timed separately, with the bytes each moves
#   hbm_read_probe.txt   what a pure read stream of the LS voter's 708 MB achieves on the box (the voter's ceiling)
#   monitor_cost.txt     the forward with and without the always-armed f16x2 range monitor, alternating, one call
#   head_probe.txt       the training heads' kernels alone: partly written records against dense rows and whole records
#   grad_ranges.txt      max |dY| per layer over six training steps (what the loss exponent has to span)
#   train_backward_ab.txt  the training step with the backward on fp16 pairs / on the exact split, alternating, one call
set -u
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_probes; rm -rf $O; mkdir -p $O; cd $R
STAMP="# stamp $(python3 bench.py --stamp)   ($(date -u +%Y-%m-%dT%H:%MZ), tools/probes_r06.sh)"
{ echo "$STAMP"; echo "# tools/debug/hs_trace.py (HS_TRACE variant of conv_hsplit): block 0, consumer wave 0 / loader wave 4; per barrier phase 'tag:cycles since the previous stamp'"
  echo "# consumers: 2 slice MFMA loop done, 3 image block done, 4 epilogue done, 0 barrier arrival, 1 release; loaders: 2 register -> LDS stores done, 3 requests done, 4 interpolation done, 5 twin epilogue done"
  CASAPOSE_HIP_LIB=$R/variants/lib_HS_TRACE.so HS_TRACE_ROWS=9 python3 tools/debug/hs_trace.py stage1_unit1_conv1 stage1_unit1_conv2 stage2_unit2_conv1 pv_block_3_conv2d pv_block_4_conv2d pv_block_5_conv2d pv_block_6_prepare_conv2d pv_block_8_prepare_conv2d pv_block_9_prepare_conv2d pv_block_10_prepare_conv2d 2>&1 | grep -v amdgpu.ids
} > $O/hsplit_trace.txt
{ echo "$STAMP"; python3 tools/debug/wino_parts.py 2>&1 | grep -v amdgpu.ids
  echo "# decoder block 2 on the Winograd path again (CASAPOSE_WINO_DIRECT_128=0) against the direct kernel (default), per-layer table rows:"
  CASAPOSE_WINO_DIRECT_128=0 python3 tools/layer_times.py --reps 10 2>/dev/null | grep "pv_block_2_\|stage2_unit2_conv1\|whole"
  python3 tools/layer_times.py --reps 10 2>/dev/null | grep "pv_block_2_\|stage2_unit2_conv1\|whole"
} > $O/wino_parts.txt
{ echo "$STAMP"; python3 tools/debug/hbm_read_probe.py 2>&1 | grep -v amdgpu.ids
  echo "# the voter itself (bench.py --mode vote):"
  python3 bench.py --mode vote --steps 20 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({k: d[k] for k in ('value','ms_per_step','roofline') if k in d}))"
} > $O/hbm_read_probe.txt
{ echo "$STAMP"; echo "# bash tools/debug/ab_bench.sh: forward + component filter + LS voting, 30 steps; CASAPOSE_F16X2_MONITOR=0 = calibration only, '-' = every forward armed (default)"
  bash tools/debug/ab_bench.sh CASAPOSE_F16X2_MONITOR=0 - CASAPOSE_F16X2_MONITOR=0 - CASAPOSE_F16X2_MONITOR=0 - 2>&1
} > $O/monitor_cost.txt
{ echo "$STAMP"; echo "# tools/debug/head_probe.py: the training step's 1x1 head kernels alone at bs 32 / 448 x 448 (partly written 36-float records against dense rows and whole records)"
  python3 tools/debug/head_probe.py 2>&1 | grep -v amdgpu.ids
  echo "# inference: whole output records from block 10's fused head (CASAPOSE_INFER_HEAD_RECORDS=1, opt-in) against the two slice writers ('-', default), alternating, one call"
  bash tools/debug/ab_bench.sh CASAPOSE_INFER_HEAD_RECORDS=1 - CASAPOSE_INFER_HEAD_RECORDS=1 - CASAPOSE_INFER_HEAD_RECORDS=1 - 2>&1
  CASAPOSE_INFER_HEAD_RECORDS=1 python3 tools/layer_times.py --reps 10 2>/dev/null | grep -E "pv_block_(5|10)_|whole"
  python3 tools/layer_times.py --reps 10 2>/dev/null | grep -E "pv_block_(5|10)_|whole"
} > $O/head_probe.txt
{ echo "$STAMP"; echo "# tools/debug/grad_ranges.py: max |dY| per convolution op over six training steps (bs 8, 448 x 448, random initialisation)"
  python3 tools/debug/grad_ranges.py 6 2>&1 | grep -v amdgpu.ids
} > $O/grad_ranges.txt
{ echo "$STAMP"; echo "# bash tools/debug/train_ab.sh: the training step (bs 32, 448 x 448, 8 timed steps) per backward arithmetic, alternating, one call: images/s, ms per step"
  for v in "-" "CASAPOSE_TRAIN_BWD=split" "-" "CASAPOSE_TRAIN_BWD=split" "CASAPOSE_TRAIN_FWD=split CASAPOSE_TRAIN_BWD=split" "CASAPOSE_HEAD_RECORDS=0"; do
    ( if [ "$v" != "-" ]; then export $v; fi
      echo -n "[$v] "; python3 bench.py --mode train --steps 8 --warmup 3 2>/dev/null | python3 -c "import json,sys; p=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(p['value'], p['ms_per_step'], p['f16x2_backward'])" )
  done
} > $O/train_backward_ab.txt
ls -la $O; head -5 $O/wino_parts.txt
