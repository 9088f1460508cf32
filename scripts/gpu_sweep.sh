cd $GRAFT_REPO_ROOT
t() { env "$1" python bench.py --cpu-scenes 0 --no-roofline 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys; d=json.load(open('/tmp/o.json')); print(sys.argv[1], d['value'], d['ms_per_step'])" $1; }
t DET6D_FPS_HOG_KB=0; t DET6D_FPS_HOG_KB=120; t DET6D_FPS_HOG_KB=60; t DET6D_FPS_HOG_KB=0
