cd $GRAFT_REPO_ROOT
DET6D_FPS_DBG=10 timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fps or skip or sampler" 2>&1 | tail -2
t() { python bench.py --cpu-scenes 0 $* 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys,os; d=json.load(open('/tmp/o.json')); print('dbg', os.environ.get('DET6D_FPS_DBG'), d['value'], d['ms_per_step'], 'fps alone ms', d['index_kernels']['fps_sa1_ms'])" $*; }
t; DET6D_FPS_DBG=10 t; t; DET6D_FPS_DBG=10 t
