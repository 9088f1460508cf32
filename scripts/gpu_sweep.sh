cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_compact_gpu.py -x -q -m gpu 2>&1 | tail -3
t() { python bench.py --cpu-scenes 0 --no-roofline 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys,os; d=json.load(open('/tmp/o.json')); print('smin', os.environ.get('DET6D_COMPACT_SMIN'), 'split', os.environ.get('DET6D_COMPACT_SPLIT'), d['value'], d['ms_per_step'])"; }
for r in 1 2; do DET6D_COMPACT_SMIN=1 DET6D_COMPACT_SPLIT=4 t; DET6D_COMPACT_SMIN=1 DET6D_COMPACT_SPLIT=1 t; DET6D_COMPACT_SMIN=2 DET6D_COMPACT_SPLIT=4 t; DET6D_COMPACT_SMIN=1 DET6D_COMPACT_SPLIT=2 t; DET6D_COMPACT_SMIN=1 DET6D_COMPACT_SPLIT=8 t; done
python scripts/gpu_linear_breakdown.py 2>/dev/null | tail -42
