cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 900 python bench.py > gpurun_out/bench_r01_s.json 2> gpurun_out/bench_r01_s.err
tail -1 gpurun_out/bench_r01_s.json | cut -c1-3000
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
