cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], "cold", d["cold"]["scenes_per_s"], d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --no-roofline --steps 20 --warmup 5"
python3 bench.py $B --cfg slopedkitti_models/det6d_car.yaml --tilt 2>/dev/null | python3 -c "$show" alone-sloped
python3 bench.py $B --cfg kitti_models/det6d_3class.yaml --batch 4 2>/dev/null | python3 -c "$show" alone-3class
python3 - <<'PY' &
import os, time, torch
ss = [torch.cuda.Stream() for _ in range(22)]
x = torch.zeros(1024, device='cuda')
for s in ss:
    with torch.cuda.stream(s):
        x.add_(1)
torch.cuda.synchronize()
time.sleep(60)
PY
sleep 8
python3 bench.py $B --cfg slopedkitti_models/det6d_car.yaml --tilt 2>/dev/null | python3 -c "$show" beside-idle-22-streams-sloped
python3 bench.py $B --cfg kitti_models/det6d_3class.yaml --batch 4 2>/dev/null | python3 -c "$show" beside-idle-22-streams-3class
wait
