cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["ms_per_step"], "cold", d["cold"]["scenes_per_s"], "lat", d.get("latency"), d["selfcheck"], [ (g["group"], g["fill"]) for g in d.get("compact_fill", [])])'
B="--no-roofline --no-legs --cpu-scenes 0 --steps 64 --warmup 16"
( time python3 bench.py $B --cfg slopedkitti_models/det6d_car.yaml --tilt 2>&1 | python3 -c "$show" cfg3-sloped ) 2>&1 | grep -v "^$\|user\|sys"
( time python3 bench.py $B --cfg kitti_models/det6d_3class.yaml --batch 4 2>&1 | python3 -c "$show" cfg4-3class-b4 ) 2>&1 | grep -v "^$\|user\|sys"
( time python3 bench.py $B --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8 2>&1 | python3 -c "$show" cfg5-65536 ) 2>&1 | grep -v "^$\|user\|sys"
( time python3 bench.py $B --scene beam 2>&1 | python3 -c "$show" beam ) 2>&1 | grep -v "^$\|user\|sys"
( time python3 bench.py $B --scene beam --tilt --cfg slopedkitti_models/det6d_car.yaml 2>&1 | python3 -c "$show" beam-sloped ) 2>&1 | grep -v "^$\|user\|sys"
