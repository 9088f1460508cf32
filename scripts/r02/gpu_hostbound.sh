# is the single-process pipeline host-bound?  (two rank processes on ONE GPU reached 14.9 k scenes/s against 10.3 k)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02b
export GPU_MAX_HW_QUEUES=24
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["n_gpus"], d["value"], "cold", d["cold"]["scenes_per_s"], "blocked", d["host_blocked_frac"], d["per_rank_scenes_per_s"], d["selfcheck"])'
B="--no-roofline --no-legs --cpu-scenes 0"
for pr in 8 8 8 8; do python3 bench.py --steps 20 --warmup 5 $B --preroll $pr 2>/dev/null | python3 -c "$show" short-preroll$pr; done
for pr in 8 8 8 8; do python3 bench.py --steps 20 --warmup 5 $B --preroll $pr 2>/dev/null | python3 -c "$show" short-preroll$pr; done
python3 bench.py --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" long
python3 bench.py --steps 960 --warmup 48 $B 2>/dev/null | python3 -c "$show" longer
export DET6D_BENCH_BACKEND=gloo
python3 bench.py --gpus 2 --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" 2proc-16+6
python3 bench.py --gpus 3 --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" 3proc-16+6
GPU_MAX_HW_QUEUES=12 python3 bench.py --gpus 2 --steps 192 --warmup 48 $B --streams 8 --sampler-streams 3 2>/dev/null | python3 -c "$show" 2proc-8+3-q12
GPU_MAX_HW_QUEUES=16 python3 bench.py --gpus 2 --steps 192 --warmup 48 $B --streams 12 --sampler-streams 4 2>/dev/null | python3 -c "$show" 2proc-12+4-q16
python3 bench.py --gpus 4 --steps 192 --warmup 48 $B --streams 8 --sampler-streams 3 2>/dev/null | python3 -c "$show" 4proc-8+3
