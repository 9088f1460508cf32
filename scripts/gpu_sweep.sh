cd $GRAFT_REPO_ROOT
t() { env "$1" python bench.py --cpu-scenes 0 --no-roofline --streams $2 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys; d=json.load(open('/tmp/o.json')); print(sys.argv[1], 'streams', sys.argv[2], d['value'], d['ms_per_step'])" $1 $2; }
t DET6D_FPS_SKIP=16 14; t DET6D_FPS_SKIP=8 14; t DET6D_FPS_SKIP=0 14; t DET6D_FPS_SKIP=8 15; t DET6D_FPS_SKIP=16 15; t DET6D_FPS_SKIP=16 13
