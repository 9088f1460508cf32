cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-scenes 0 --no-roofline 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys; d=json.load(open('/tmp/o.json')); print(sys.argv[1], d['value'], d['ms_per_step'])" "$1"; }
DET6D_FPS_SKIP=16 python scripts/gpu_fps_cells.py 2>&1 | head -3
DET6D_FPS_SKIP=16 run skip16
run skip8
DET6D_FPS_SKIP=0 run fat
