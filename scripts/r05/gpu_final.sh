# final artifacts of round 5: the whole GPU suite, the driver's command (every leg), PMC + kernel stats on the three workloads, the
# PMC pass over the PIPELINED worker (steady-state slice), pipelined traces with the overlap analysis, --gpus 2 dry run
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r05_final
( time timeout 2400 python3 -m pytest tests -m gpu -q ) > gpurun_out/r05_final/pytest.log 2>&1; tail -4 gpurun_out/r05_final/pytest.log | cut -c1-200
export GPU_MAX_HW_QUEUES=24
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_final/bench_20.log 2> gpurun_out/r05_final/bench_20.err
grep '^{' gpurun_out/r05_final/bench_20.log | cut -c1-200; tail -4 gpurun_out/r05_final/bench_20.err
bash scripts/r05/gpu_pmc_all.sh z
bash scripts/r05/gpu_final_pipe.sh
cd $GRAFT_REPO_ROOT
DET6D_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-legs > gpurun_out/r05_final/bench_2ranks.log 2> gpurun_out/r05_final/bench_2ranks.err
python3 - <<'PY'
import json
for l in open('gpurun_out/r05_final/bench_2ranks.log'):
    if l.startswith('{'):
        d = json.loads(l); print('2 ranks on one GPU (gloo dry run):', d['n_gpus'], d['value'], d['per_rank_scenes_per_s'], d['selfcheck'], d['ranks_seen'])
PY
find gpurun_out -name "*.csv" -size +3M -delete
