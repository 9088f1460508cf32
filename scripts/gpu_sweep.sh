cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-scenes 0 --no-roofline 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys; d=json.load(open('/tmp/o.json')); print(sys.argv[1], d['value'], d['ms_per_step'])" "$1"; }
run base
cp de6d_amd/csrc/libdet6d_hip.so /tmp/base.so
cp de6d_amd/csrc/libdet6d_hip_prio.so de6d_amd/csrc/libdet6d_hip.so
run gemm_prio3
cp /tmp/base.so de6d_amd/csrc/libdet6d_hip.so
