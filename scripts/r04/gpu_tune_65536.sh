#!/bin/bash
cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
mkdir -p gpurun_out/r04
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], "cold", d["cold"]["scenes_per_s"], "p50/p99", d.get("latency_under_load",{}).get("ms_p50_p99"), d["selfcheck"])'
B="--cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8 --steps 40 --warmup 8 --no-legs --cpu-scenes 0 --no-roofline"
while read -r cfg; do
python3 bench.py $B $cfg 2>/dev/null | python3 -c "$show" "$cfg"
done <<'CFGS' 2>&1 | tee gpurun_out/r04/tune_65536b.log
--streams 8 --group 2 --prefetch 2
--streams 4 --group 2 --prefetch 2
--streams 4 --group 1 --prefetch 2
--merge 1 --streams 16 --group 4 --prefetch 2
--merge 1 --streams 8 --group 4 --prefetch 2
--merge 2 --streams 8 --group 2 --prefetch 2
--merge 2 --streams 4 --group 2 --prefetch 1
CFGS
