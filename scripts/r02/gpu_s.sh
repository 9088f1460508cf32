cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --worker --no-roofline --steps 192 --warmup 48"
for i in 1 2; do
python3 bench.py $B 2>/dev/null | python3 -c "$show" all-fused
DET6D_TMP_NO_HEADB=1 python3 bench.py $B 2>/dev/null | python3 -c "$show" headB-3launch
DET6D_TMP_ONLY_HEADB=1 python3 bench.py $B 2>/dev/null | python3 -c "$show" only-headB-fused
python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" beam-all-fused
DET6D_TMP_NO_HEADB=1 python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" beam-headB-3launch
done
