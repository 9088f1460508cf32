# streaming form of the wide grouped MLP: parity, then bench with the form off / on / on for both head groups
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_compact_gpu.py -x -q -m gpu -k "group_kernel" 2>&1 | tail -3
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print(sys.argv[1], d["value"], d["selfcheck"], "roof", r["frac"], r["kernel_ms_per_pass"], "sat", r["saturated"]["frac"], r["saturated"]["family_ms_per_pass"]); print([ (x[0], x[3], x[4]) for x in r["launches"] if x[3] > 100])'
for w in 0 1 2; do
DET6D_GROUP_STREAM=$w python3 bench.py --steps 20 --warmup 5 --no-legs --cpu-scenes 0 2>/dev/null | python3 -c "$show" "stream=$w"
DET6D_GROUP_STREAM=$w python3 bench.py --steps 20 --warmup 5 --no-legs --cpu-scenes 0 --scene beam 2>/dev/null | python3 -c "$show" "beam stream=$w"
done
