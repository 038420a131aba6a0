#!/bin/bash
# ONE GPU-box call that produces every artefact of a round's profile set from the SAME tree, stamped (python3 bench.py --stamp):
#   gpurun --timeout 3600 -- 'bash tools/profile_round.sh r03 [suites]'   ->   gpurun_out/<tag>/...   ->   python tools/make_summary.py r03
# Pieces (each its own process; the --pmc passes use --kernel-trace only, as the pool requires):
#   bench.json                 default `python3 bench.py` (headline + roofline + fp32-MFMA line + training leg + CPU baseline)
#   bench_bs32.json            the same at bs 32 (the batch the north star quotes its MFMA target on), no CPU baseline / training leg
#   bench_f32.json             CASAPOSE_INFER_CONV_MODE=f32 (fp32 MFMA everywhere)      bench_split.json  =split (exact bf16 splits)      bench_bf16.json  =bf16
#   trace/                     rocprofv3 --kernel-trace --stats of the default bench (kernel stats csv)
#   pmc_traffic/               FETCH_SIZE and WRITE_SIZE in two --pmc passes -> traffic.json / summary.txt
#   pmc_mfma/                  SQ MFMA-busy counters -> summary.txt
#   layer_times*.txt           per-layer table, default (f16x2), split and f32
#   bench_train*.json, train_trace/, train_times.txt      the training step (default, CASAPOSE_CONV_MODE=f32 / bf16)
#   bench_vote.json, vote_trace/                           the voting stage alone
set -u
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun (GRAFT_REPO_ROOT = root of the snapshot)}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-r03}
O=$R/gpurun_out/$tag
rm -rf $O && mkdir -p $O
python3 $R/bench.py --stamp > $O/stamp.json
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python3 $R/bench.py --steps 20 --warmup 5 --batch 32 --no-cpu-baseline --no-train-leg > $O/bench_bs32.json 2>> $O/bench.err
CASAPOSE_INFER_CONV_MODE=f32 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-optin > $O/bench_f32.json 2>> $O/bench.err
CASAPOSE_INFER_CONV_MODE=split python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-optin > $O/bench_split.json 2>> $O/bench.err
CASAPOSE_INFER_CONV_MODE=bf16 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-optin > $O/bench_bf16.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-optin > $O/bench_profiled.json 2> $O/rocprof.err
mkdir -p $O/pmc_traffic $O/pmc_mfma
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_traffic/fetch -o fetch --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-train-leg --no-optin > $O/pmc_traffic/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_traffic/write -o write --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-train-leg --no-optin > $O/pmc_traffic/write.log 2>&1
python3 $R/tools/pmc_traffic_parse.py $O/pmc_traffic > $O/pmc_traffic/summary.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma/p -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-train-leg --no-optin > $O/pmc_mfma/run.log 2>&1
python3 $R/tools/pmc_mfma_parse.py $O/pmc_mfma > $O/pmc_mfma/summary.txt 2>&1
python3 $R/tools/layer_times.py > $O/layer_times.txt 2>&1
CASAPOSE_INFER_CONV_MODE=f32 python3 $R/tools/layer_times.py > $O/layer_times_f32.txt 2>&1
CASAPOSE_INFER_CONV_MODE=split python3 $R/tools/layer_times.py > $O/layer_times_split.txt 2>&1
python3 $R/bench.py --mode train --steps 8 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err
CASAPOSE_CONV_MODE=f32 CASAPOSE_WINO_GEMM=f32 python3 $R/bench.py --mode train --steps 8 --warmup 3 > $O/bench_train_f32.json 2>> $O/bench_train.err
CASAPOSE_CONV_MODE=bf16 python3 $R/bench.py --mode train --steps 8 --warmup 3 > $O/bench_train_bf16.json 2>> $O/bench_train.err
CASAPOSE_TRAIN_FWD=split CASAPOSE_TRAIN_BWD=split python3 $R/bench.py --mode train --steps 8 --warmup 3 > $O/bench_train_exact.json 2>> $O/bench_train.err
rocprofv3 --kernel-trace --stats -d $O/train_trace -o train --output-format csv -- python3 $R/bench.py --mode train --steps 5 --warmup 2 > $O/bench_train_profiled.json 2> $O/rocprof_train.err
python3 $R/tools/train_times.py > $O/train_times.txt 2>&1
python3 $R/bench.py --mode vote --steps 20 --warmup 5 > $O/bench_vote.json 2> $O/bench_vote.err
rocprofv3 --kernel-trace --stats -d $O/vote_trace -o vote --output-format csv -- python3 $R/bench.py --mode vote --steps 10 --warmup 3 > $O/bench_vote_profiled.json 2> $O/rocprof_vote.err
if [ "${2:-}" = "suites" ]; then   # the -m gpu suite per conv mode, on the same box and tree (about 4 minutes each)
  cd $R
  python3 bench.py --stamp > $O/gputests_stamp.json
  python3 -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/gputests_default_f16x2.log
  CASAPOSE_INFER_CONV_MODE=split python3 -m pytest tests -q -m gpu --ignore=tests/test_gpu_train.py --ignore=tests/test_gpu_scripts.py --ignore=tests/test_gpu_dp.py 2>&1 | tail -4 > $O/gputests_infer_conv_mode_split.log
  CASAPOSE_INFER_CONV_MODE=f32 python3 -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/gputests_infer_conv_mode_f32.log
  CASAPOSE_CONV_MODE=f32 CASAPOSE_WINO_GEMM=f32 python3 -m pytest tests/test_gpu_train.py tests/test_gpu_scripts.py tests/test_gpu_dp.py -q -m gpu 2>&1 | tail -4 > $O/gputests_train_conv_mode_f32.log
  CASAPOSE_TRAIN_FWD=split CASAPOSE_TRAIN_BWD=split python3 -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -4 > $O/gputests_train_exact.log
  cd /tmp
fi
python3 $R/bench.py --stamp > $O/stamp_end.json
# keep the merge small: only the stats / counter summaries travel back, not the raw traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; ls $O; cat $O/bench.json | head -c 600
