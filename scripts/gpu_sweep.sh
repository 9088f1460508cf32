cd $GRAFT_REPO_ROOT
python bench.py --batch 8 --streams 1 --cpu-scenes 0 --no-roofline --steps 60 --warmup 10 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json; d=json.load(open('/tmp/o.json')); print('batch 8, 1 stream:', d['value'], d['ms_per_step'])"
python bench.py --batch 1 --streams 1 --cpu-scenes 0 --no-roofline --steps 100 --warmup 20 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json; d=json.load(open('/tmp/o.json')); print('batch 1, 1 stream:', d['value'], d['ms_per_step'])"
python bench.py --h2d --cpu-scenes 0 --no-roofline 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json; d=json.load(open('/tmp/o.json')); print('h2d:', d['value'], d['ms_per_step'])"
