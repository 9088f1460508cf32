#!/bin/bash
# pipeline shape after the sampler change: sampler streams x prefetch x main streams (benchmark scenes, then ray-cast for the best few)
cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
mkdir -p gpurun_out/r04
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], "cold", d["cold"]["scenes_per_s"], d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --no-roofline --steps 20 --warmup 5"
for ss in 4 6 8; do for pf in 3 4 6; do
python3 bench.py $B --sampler-streams $ss --prefetch $pf 2>/dev/null | python3 -c "$show" "uniform ss=$ss pf=$pf"
done; done 2>&1 | tee gpurun_out/r04/tune_pipe.log
for st in 12 14 18; do
python3 bench.py $B --streams $st 2>/dev/null | python3 -c "$show" "uniform streams=$st"
done 2>&1 | tee -a gpurun_out/r04/tune_pipe.log
