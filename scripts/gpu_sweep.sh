cd $GRAFT_REPO_ROOT
t() { python bench.py --cpu-scenes 0 --no-roofline $* 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys,os; d=json.load(open('/tmp/o.json')); print(sys.argv[1:], d['value'], d['ms_per_step'])" $*; }
t --batch 32 --group 1 --streams 8 --prefetch 2 --sampler-streams 4 --steps 48 --warmup 16
t --batch 32 --group 1 --streams 12 --prefetch 3 --sampler-streams 4 --steps 48 --warmup 16
t --batch 16 --group 2 --streams 12 --prefetch 3 --sampler-streams 4 --steps 96 --warmup 24
t --batch 4 --group 8 --streams 16 --prefetch 2 --sampler-streams 4 --steps 384 --warmup 96
