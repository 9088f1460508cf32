# timing experiment (experiments build, WRONG results with the flag on): the grouped-MLP kernel with its weight stream served
# from the CU's cache instead of L2
cd $GRAFT_REPO_ROOT
export DET6D_EXPERIMENTS_LIB=1
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print(sys.argv[1], d["value"], "roof", r["frac"], r["kernel_ms_per_pass"], "sat", r["saturated"]["frac"], r["saturated"]["family_ms_per_pass"]); print([ (x[0], x[3], x[4]) for x in r["launches"] if x[3] > 100])'
for w in 0 1; do
DET6D_GROUP_WHATIF=$w python3 bench.py --steps 20 --warmup 5 --no-legs --cpu-scenes 0 2>/dev/null | python3 -c "$show" "whatif=$w"
done
