cd $GRAFT_REPO_ROOT
t() { env "$1" python bench.py --cpu-scenes 0 --no-roofline --streams $2 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys; d=json.load(open('/tmp/o.json')); print(sys.argv[1], 'streams', sys.argv[2], d['value'], d['ms_per_step'])" $1 $2; }
t X=1 13; t X=1 14; t X=1 15; t DET6D_FORKED_SAMPLERS=1 15; t X=1 15
python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -1
python bench.py --batch 8 --streams 1 --cpu-scenes 0 --no-roofline --steps 60 --warmup 10 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json; d=json.load(open('/tmp/o.json')); print('batch 8, 1 stream (sequential samplers):', d['ms_per_step'])"
