set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04u; mkdir -p $O
run() { # name args
  for v in base HS_NOLOAD HS_NOWLOAD HS_NOBOTH; do
    if [ $v = base ]; then unset CASAPOSE_HIP_LIB; else export CASAPOSE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_$v.so; fi
    echo -n "$1 $v: "; python tools/one_layer.py ${@:2} --reps 20 2>/dev/null | tail -n 1
  done
  unset CASAPOSE_HIP_LIB
}
for tile in 102; do
run "b4 t$tile" --batch 16 --h 240 --w 320 --cin 128 --cout 32 --dil 1 --tile $tile
run "b5 t$tile" --batch 16 --h 480 --w 640 --cin 32 --cout 32 --dil 1 --tile $tile
run "b6 t$tile" --batch 16 --h 60 --w 80 --cin 512 --cout 64 --dil 1 --tile $tile
run "b3 t$tile" --batch 16 --h 120 --w 160 --cin 192 --cout 64 --dil 1 --tile $tile
done 2>&1 | tee $O/bound2.txt
