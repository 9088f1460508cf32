cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_compact_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -4
t() { python bench.py --cpu-scenes 0 --no-roofline $* 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys,os; d=json.load(open('/tmp/o.json')); print(sys.argv[1:], d['value'], d['ms_per_step'])" $*; }
t; t; t --streams 14 --prefetch 4; t --streams 16 --prefetch 3
