cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-scenes 0 --no-roofline 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys; d=json.load(open('/tmp/o.json')); print(sys.argv[1], d['value'], d['ms_per_step'])" "$1"; }
run waves_per_eu3_a
run waves_per_eu3_b
