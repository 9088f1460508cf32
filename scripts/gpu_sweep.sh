cd $GRAFT_REPO_ROOT
t() { GPU_MAX_HW_QUEUES=$1 python bench.py --cpu-scenes 0 --no-roofline --streams $2 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys; d=json.load(open('/tmp/o.json')); print('queues', sys.argv[1], 'streams', sys.argv[2], d['value'], d['ms_per_step'])" $1 $2; }
t 24 18; t 24 20; t 24 22; t 24 23; t 28 24; t 28 26; t 24 20
