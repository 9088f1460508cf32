#!/bin/bash
# same-box, SAME-LIBRARY A/B in the pipeline (experiments build on both sides): multi-pick sampler (DET6D_FPS_SEQ=1) vs wave-skip
cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24 DET6D_EXPERIMENTS_LIB=1
mkdir -p gpurun_out/r04
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], "cold", d["cold"]["scenes_per_s"], "lat", d.get("latency",{}).get("ms_per_batch"), "lat_b1", d.get("latency_b1", {}).get("ms_per_frame"), d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --no-roofline --steps 20 --warmup 5"
for i in 1 2; do
DET6D_FPS_SEQ=1 python3 bench.py $B 2>/dev/null | python3 -c "$show" "uniform seq"
DET6D_FPS_SEQ=0 python3 bench.py $B 2>/dev/null | python3 -c "$show" "uniform skip"
DET6D_FPS_SEQ=1 python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" "beam seq"
DET6D_FPS_SEQ=0 python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" "beam skip"
done 2>&1 | tee gpurun_out/r04/ab_fps2.log
