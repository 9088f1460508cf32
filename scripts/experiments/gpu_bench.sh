set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 60 --warmup 8 2>&1 | tail -2 | tee gpurun_out/bench_r01.log
for s in 1 2 8; do python bench.py --steps 60 --warmup 8 --streams $s --cpu-scenes 0 --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['streams'], d['value'], d['ms_per_step'])"; done
