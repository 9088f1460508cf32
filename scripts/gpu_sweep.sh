cd $GRAFT_REPO_ROOT
t() { python scripts/gpu_whatif2.py --cpu-scenes 0 --no-roofline $* 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys,os; d=json.load(open('/tmp/o.json')); print(os.environ.get('WHATIF'), sys.argv[1:], d['value'], d['ms_per_step'])" $*; }
WHATIF=base t; WHATIF=halfk t; WHATIF=nofps,halfk t; WHATIF=nofps t
