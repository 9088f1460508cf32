# pipeline shape for coalesced passes (a step stays a batch of 8; --merge batches share a pass)
cd $GRAFT_REPO_ROOT
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], "mean", d["config"].get("window_ms_mean"), "cold", d["cold"]["scenes_per_s"], d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --no-roofline"
run() { python3 bench.py $B "$@" 2>&1 | grep -v "^selfcheck" | python3 -c "$show" "$*"; }
run --steps 20 --warmup 5
run --steps 20 --warmup 5 --streams 16
run --steps 20 --warmup 5 --streams 16 --prefetch 6
run --steps 20 --warmup 5 --streams 16 --prefetch 8
run --steps 20 --warmup 5 --streams 18 --prefetch 4
run --steps 20 --warmup 5 --streams 18 --prefetch 6
run --steps 20 --warmup 5 --streams 14 --prefetch 6
run --steps 20 --warmup 5 --streams 12 --prefetch 8
run --steps 20 --warmup 5 --streams 16 --prefetch 6 --sampler-streams 4
run --steps 20 --warmup 5 --streams 16 --prefetch 6 --sampler-streams 8
run --steps 20 --warmup 5 --streams 20 --prefetch 6 --sampler-streams 4
run --steps 20 --warmup 5 --streams 16 --prefetch 6 --scene beam
run --steps 20 --warmup 5 --cfg kitti_models/det6d_3class.yaml --batch 4
run --steps 20 --warmup 5 --cfg kitti_models/det6d_3class.yaml --batch 4 --merge 1
run --steps 20 --warmup 5 --cfg synthetic_models/det6d_65536.yaml --points 65536
run --steps 20 --warmup 5 --cfg synthetic_models/det6d_65536.yaml --points 65536 --merge 1
