cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t12; mkdir -p $out
run() { tag=$1; shift
  timeout 600 python3 scripts/r05/whatif_occupancy.py "$@" --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline > $out/w_$tag.log 2> $out/w_$tag.err
  grep '^{' $out/w_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('occupier $tag:', d['value'], d['selfcheck'])" || tail -3 $out/w_$tag.err; }
run none 0 0 0
run r96_lds34k 96 34816 2000
run r80_lds98k 80 100352 2000
run r80_lds34k 80 34816 2000
run r56_lds34k 56 34816 2000
run r40_lds34k 40 34816 2000
run r16_lds34k 16 34816 2000
run r96_1ms 96 34816 1000
