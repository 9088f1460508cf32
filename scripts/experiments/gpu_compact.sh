cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_compact_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -15
for s in 16 22; do
  timeout 300 python bench.py --cpu-scenes 0 --no-roofline --streams $s 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('compact streams', d['config']['streams'], d['value'], d['ms_per_step'])"
done
DET6D_COMPACT_NO_CHAIN=1 timeout 300 python bench.py --cpu-scenes 0 --no-roofline --streams 22 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('compact nochain streams', d['config']['streams'], d['value'], d['ms_per_step'])"
