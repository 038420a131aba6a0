set -u
cd $GRAFT_REPO_ROOT
bash tools/pmc_one_layer.sh h2_b4 --batch 16 --h 240 --w 320 --cin 128 --cout 32 --dil 1 --tile 102 > /dev/null 2>&1
bash tools/pmc_one_layer.sh h2_b5 --batch 16 --h 480 --w 640 --cin 32 --cout 32 --dil 1 --tile 102 > /dev/null 2>&1
bash tools/pmc_one_layer.sh h2_b6 --batch 16 --h 60 --w 80 --cin 512 --cout 64 --dil 1 --tile 102 > /dev/null 2>&1
bash tools/pmc_one_layer.sh h2_s1 --batch 16 --h 120 --w 160 --cin 64 --cout 64 --dil 1 --tile 102 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for t in h2_b4 h2_b5 h2_b6 h2_s1; do echo "== $t"; python tools/pmc_parse.py gpurun_out/pmc_$t conv_hsplit; done > gpurun_out/pmc_hsplit_f16x2_summary.txt 2>&1
cat gpurun_out/pmc_hsplit_f16x2_summary.txt
