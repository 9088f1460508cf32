# experiments build: LDS reservation of the first-layer sampler (KB left to other workgroups on its CU) with 32-scene passes
cd $GRAFT_REPO_ROOT
export DET6D_EXPERIMENTS_LIB=1
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_mean"], "cold", d["cold"]["scenes_per_s"], d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --no-roofline --steps 20 --warmup 5"
for h in 0 16 48 96; do
DET6D_FPS_LDS_HOG=$h python3 bench.py $B 2>/dev/null | python3 -c "$show" "uniform hog=$h"
DET6D_FPS_LDS_HOG=$h python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" "beam hog=$h"
done
