cd $GRAFT_REPO_ROOT
t() { python bench.py --cpu-scenes 0 --no-roofline $* 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys,os; d=json.load(open('/tmp/o.json')); print('dbg', os.environ.get('DET6D_FPS_DBG'), sys.argv[1:], d['value'], d['ms_per_step'])" $*; }
t; DET6D_FPS_DBG=9 t; t; DET6D_FPS_DBG=9 t
