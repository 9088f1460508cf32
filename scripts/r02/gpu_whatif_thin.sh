# timing experiment (experiments build, WRONG picks): the first-layer sampler as 4 thin workgroups (4 waves, 88 VGPRs) per scene
# that leave room for the head's grouped-MLP workgroups on their CUs, vs the one 16-wave workgroup per scene
cd $GRAFT_REPO_ROOT
export DET6D_EXPERIMENTS_LIB=1
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_mean"], "cold", d["cold"]["scenes_per_s"], d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --no-roofline --steps 20 --warmup 5"
for i in 1; do
for t in 0 1; do
DET6D_FPS_THIN=$t python3 bench.py $B 2>/dev/null | python3 -c "$show" "uniform thin=$t"
DET6D_FPS_THIN=$t python3 bench.py $B --prefetch 6 2>/dev/null | python3 -c "$show" "uniform thin=$t prefetch 6"
DET6D_FPS_THIN=$t python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" "beam thin=$t"
done; done
