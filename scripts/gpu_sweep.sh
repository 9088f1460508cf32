cd $GRAFT_REPO_ROOT
for T in 1 2; do
sed -i "s/constexpr int kPackTiles = [0-9]*;/constexpr int kPackTiles = $T;/" de6d_amd/csrc/mlp_chain.hip
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
t() { python bench.py --cpu-scenes 0 --no-roofline $* 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys,os; d=json.load(open('/tmp/o.json')); print('T', sys.argv[1], d['value'], d['ms_per_step'])" $T; }
t; t
done
