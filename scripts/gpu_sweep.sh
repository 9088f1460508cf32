cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-scenes 0 --no-roofline 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys; d=json.load(open('/tmp/o.json')); print(sys.argv[1], d['value'], d['ms_per_step'])" "$1"; }
DET6D_LINEAR_FAST64=1 run fast64
run base
DET6D_LINEAR_FAST64=1 run fast64_again
