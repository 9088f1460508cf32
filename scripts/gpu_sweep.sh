cd $GRAFT_REPO_ROOT
for v in 162 16 84 8; do echo "== DET6D_FPS_SKIP=$v"; DET6D_FPS_SKIP=$v python scripts/gpu_fps_cells.py 2>&1 | grep "n=16384\|all-equal"; done
t() { env "$1" python bench.py --cpu-scenes 0 --no-roofline 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys; d=json.load(open('/tmp/o.json')); print(sys.argv[1], d['value'], d['ms_per_step'])" $1; }
t DET6D_FPS_SKIP=162; t DET6D_FPS_SKIP=16; t DET6D_FPS_SKIP=84; t DET6D_FPS_SKIP=162
