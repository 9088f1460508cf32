cd $GRAFT_REPO_ROOT
t() { GPU_MAX_HW_QUEUES=$1 python bench.py --cpu-scenes 0 --no-roofline --streams $2 --group $3 --prefetch $4 --sampler-streams $5 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys,os; d=json.load(open('/tmp/o.json')); print('queues', sys.argv[1], 'main', sys.argv[2], 'group', sys.argv[3], 'prefetch', sys.argv[4], 'samp', sys.argv[5], d['value'], d['ms_per_step'])" $1 $2 $3 $4 $5; }
t 24 16 4 4 3; t 24 18 4 4 3; t 24 18 4 4 2; t 24 17 4 4 3; t 24 16 4 4 2; t 24 19 4 4 2
