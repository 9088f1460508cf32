# final artifacts: the driver's command (every leg), then --gpus 2 on the one GPU of the box (gloo barrier: both ranks share cuda:0)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02_final
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r02_final/bench_20.log 2> gpurun_out/r02_final/bench_20.err
grep '^{' gpurun_out/r02_final/bench_20.log | cut -c1-200; tail -5 gpurun_out/r02_final/bench_20.err
DET6D_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r02_final/bench_2ranks.log 2> gpurun_out/r02_final/bench_2ranks.err
python3 - <<'PY'
import json
for l in open('gpurun_out/r02_final/bench_2ranks.log'):
    if l.startswith('{'):
        d = json.loads(l); print('2 ranks on one GPU:', d['n_gpus'], d['value'], d['per_rank_scenes_per_s'], d['selfcheck'], d['ranks_seen'])
PY
tail -3 gpurun_out/r02_final/bench_2ranks.err
