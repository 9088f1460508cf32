cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 900 python bench.py > gpurun_out/bench_r01_zz.json 2> gpurun_out/bench_r01_zz.err
tail -1 gpurun_out/bench_r01_zz.json | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/gpu_prof.sh r01z 2>&1 | tail -3
bash scripts/gpu_pmc.sh 2>&1 | tail -3
