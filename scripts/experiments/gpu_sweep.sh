cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_compact_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -2
t() { python bench.py --cpu-scenes 0 --no-roofline $* 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys,os; d=json.load(open('/tmp/o.json')); print(sys.argv[1:], d['value'], d['ms_per_step'])" $*; }
t; t
bash scripts/gpu_launch_list.sh > /dev/null 2>&1; grep -n "compact_place" gpurun_out/launch_list/one_pass.txt | cut -c1-100
