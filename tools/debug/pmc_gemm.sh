set -u
export CASAPOSE_GEMM_ONE=${CASAPOSE_GEMM_ONE:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_gemm; rm -rf $O; mkdir -p $O
for shape in "512 512" "256 256" "128 128"; do
  tag=$(echo $shape | tr ' ' 'x')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/$tag/p1 -o p1 --output-format csv -- python3 $R/tools/debug/gemm_one.py $shape 4 > $O/$tag.p1.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace -d $O/$tag/p2 -o p2 --output-format csv -- python3 $R/tools/debug/gemm_one.py $shape 4 > $O/$tag.p2.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_MISC --kernel-trace -d $O/$tag/p3 -o p3 --output-format csv -- python3 $R/tools/debug/gemm_one.py $shape 4 > $O/$tag.p3.log 2>&1
  echo "plain" > $O/$tag/plain.log
  echo "== $tag" >> $O/summary.txt
  python3 $R/tools/pmc_parse.py $O/$tag wino_gemm_ >> $O/summary.txt 2>&1
  python3 - >> $O/summary.txt 2>&1 <<PY
import csv, collections
agg = collections.defaultdict(list)
try:
    for r in csv.DictReader(open("$O/$tag/p3/p3_counter_collection.csv")):
        if "wino_gemm_" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print("  %-24s %.4g" % (k, sum(v[1:]) / max(len(v[1:]), 1)))
except Exception as e: print("p3:", e)
PY
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
cat $O/summary.txt
