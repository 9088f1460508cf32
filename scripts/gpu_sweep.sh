cd $GRAFT_REPO_ROOT
t() { python bench.py --cpu-scenes 0 --no-roofline $* 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys,os; d=json.load(open('/tmp/o.json')); print('est', os.environ.get('DET6D_COMPACT_ROWS_EST'), sys.argv[1:], d['value'], d['ms_per_step'])" $*; }
for e in 1 3 5 1 3; do DET6D_COMPACT_ROWS_EST=$e t; done
for e in 1 3 5; do echo est $e; DET6D_COMPACT_ROWS_EST=$e python scripts/gpu_linear_breakdown.py 2>/dev/null | tail -1; done
DET6D_COMPACT_ROWS_EST=3 python scripts/gpu_linear_breakdown.py 2>/dev/null | tail -22 | head -18
