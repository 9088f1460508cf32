# final artifacts of round 6: the whole GPU suite, the driver's command (every leg), PMC + kernel stats on the three workloads, the
# PMC pass over the PIPELINED worker (steady-state slice), pipelined traces with the overlap analysis, per-kernel clock / power,
# --gpus 2 and --gpus 8 dry runs of the entry point on the one GPU (gloo; review item 8)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r06_final
( time timeout 2700 python3 -m pytest tests -m gpu -q ) > gpurun_out/r06_final/pytest.log 2>&1; tail -4 gpurun_out/r06_final/pytest.log | cut -c1-200
export GPU_MAX_HW_QUEUES=24
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06_final/bench_20.log 2> gpurun_out/r06_final/bench_20.err
grep '^{' gpurun_out/r06_final/bench_20.log | cut -c1-200; tail -4 gpurun_out/r06_final/bench_20.err
python3 scripts/r06/kernel_power.py uniform 1.0 > gpurun_out/r06_final/kernel_power_uniform.txt 2>&1; tail -18 gpurun_out/r06_final/kernel_power_uniform.txt
python3 scripts/r06/kernel_power.py beam 0.5 > gpurun_out/r06_final/kernel_power_beam.txt 2>&1
bash scripts/r06/gpu_pmc_all.sh z
bash scripts/r06/gpu_final_pipe.sh
cd $GRAFT_REPO_ROOT
for n in 2 8; do
DET6D_BENCH_BACKEND=gloo python3 bench.py --gpus $n --steps 20 --warmup 5 --no-legs > gpurun_out/r06_final/bench_${n}ranks.log 2> gpurun_out/r06_final/bench_${n}ranks.err
python3 - $n <<'PY'
import json, sys
n = sys.argv[1]
for l in open('gpurun_out/r06_final/bench_%sranks.log' % n):
    if l.startswith('{'):
        d = json.loads(l); print(n, 'ranks on one GPU (gloo dry run):', d['n_gpus'], d['value'], d['per_rank_scenes_per_s'], d['selfcheck'], d['ranks_seen'])
PY
tail -2 gpurun_out/r06_final/bench_${n}ranks.err
done
find gpurun_out -name "*.csv" -size +3M -delete
