cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_compact_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3
t() { python bench.py --cpu-scenes 0 $* 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys,os; d=json.load(open('/tmp/o.json')); r=d['roofline']; print('griddiv', os.environ.get('DET6D_COMPACT_GRID_DIV'), d['value'], d['ms_per_step'], 'standalone', r['issued_tflops'], r['kernel_ms_per_step'], 'saturated issued TF', r['saturated']['issued_tflops'], r['saturated']['family_ms_per_pass'])" $*; }
DET6D_COMPACT_GRID_DIV=1 t; DET6D_COMPACT_GRID_DIV=3 t; DET6D_COMPACT_GRID_DIV=4 t; DET6D_COMPACT_GRID_DIV=6 t; DET6D_COMPACT_GRID_DIV=8 t; DET6D_COMPACT_GRID_DIV=1 t
