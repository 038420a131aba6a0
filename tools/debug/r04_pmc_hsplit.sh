set -u
cd $GRAFT_REPO_ROOT
bash tools/pmc_one_layer.sh hs_b4 --batch 16 --h 240 --w 320 --cin 128 --cout 32 --dil 1 --tile 100 > /dev/null 2>&1
bash tools/pmc_one_layer.sh hs_b5 --batch 16 --h 480 --w 640 --cin 32 --cout 32 --dil 1 --tile 100 > /dev/null 2>&1
bash tools/pmc_one_layer.sh hs_b6 --batch 16 --h 60 --w 80 --cin 512 --cout 64 --dil 1 --tile 100 > /dev/null 2>&1
bash tools/pmc_one_layer.sh hs_s1 --batch 16 --h 120 --w 160 --cin 64 --cout 64 --dil 1 --tile 100 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for t in hs_b4 hs_b5 hs_b6 hs_s1; do echo "== $t"; python tools/pmc_parse.py gpurun_out/pmc_$t conv_hsplit; done > gpurun_out/pmc_hsplit_summary.txt 2>&1
cat gpurun_out/pmc_hsplit_summary.txt
