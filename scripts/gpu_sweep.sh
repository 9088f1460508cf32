cd $GRAFT_REPO_ROOT
for st in 12 13 14 15 14 15; do
  python bench.py --cpu-scenes 0 --no-roofline --streams $st 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json,sys; d=json.load(open('/tmp/o.json')); print('streams', sys.argv[1], d['value'], d['ms_per_step'])" $st
done
