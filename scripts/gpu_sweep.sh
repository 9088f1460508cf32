cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_compact_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -2
t() { python bench.py --cpu-scenes 0 $* 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys,os; d=json.load(open('/tmp/o.json')); r=d['roofline']; print('k64max', os.environ.get('DET6D_LINEAR_K64MAX'), d['value'], d['ms_per_step'], 'standalone', r['achieved'], r['frac'], 'saturated', r['saturated']['tflops'], r['saturated']['family_ms_per_pass'])" $*; }
t; DET6D_LINEAR_K64MAX=0 t; t; DET6D_LINEAR_K64MAX=0 t
