cd $GRAFT_REPO_ROOT
t() { python bench.py --cpu-scenes 0 --no-roofline $* 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys,os; d=json.load(open('/tmp/o.json')); print('hog', os.environ.get('DET6D_FPS_LDS_HOG'), sys.argv[1:], d['value'], d['ms_per_step'])" $*; }
for r in 1 2; do DET6D_FPS_LDS_HOG=0 t; DET6D_FPS_LDS_HOG=8 t; DET6D_FPS_LDS_HOG=40 t; done
DET6D_FPS_LDS_HOG=0 t --streams 20; DET6D_FPS_LDS_HOG=0 t --prefetch 6 --sampler-streams 8
