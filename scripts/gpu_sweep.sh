cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_compact_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -5
t() { python bench.py --cpu-scenes 0 --no-roofline 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys,os; d=json.load(open('/tmp/o.json')); print('split', os.environ.get('DET6D_COMPACT_SPLIT'), d['value'], d['ms_per_step'])"; }
t; DET6D_COMPACT_SPLIT=0 t; t; DET6D_COMPACT_SPLIT=0 t
python scripts/gpu_linear_breakdown.py 2>/dev/null | tail -42
