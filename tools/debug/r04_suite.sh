set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1
tail -n 8 $O/suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 2 $O/smoke.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json | head -c 600; grep -o '"cpu_baseline": {[^}]*' $O/bench_default.json | head -c 600
