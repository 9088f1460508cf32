# round 6: (a) the timed-regime parity test with its failure text; (b) knobs of the group kernels at the 80-scene pass size
# (experiments build): waves per tile, streaming form, per kernel averages of eager 80-scene passes under rocprofv3 --kernel-trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r06_t3; mkdir -p $out
timeout 1700 python3 -m pytest tests/test_timed_path_gpu.py -m gpu -x -q -k "regime" > $out/regime.log 2>&1; tail -60 $out/regime.log | cut -c1-300
export DET6D_EXPERIMENTS_LIB=1
A="--steps 5 --warmup 2 --batch 80 --streams 1 --no-graph --cpu-scenes 0 --no-roofline --no-legs --preroll 0 --windows 1"
one() { tag=$1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -o k -- python3 bench.py $A > $out/$tag.log 2>&1
  f=$(find $out/$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag"; grep "mlp_group\|linear_kernel\|mlp_rows\|mlp_chain" $f | python3 -c "
import sys,csv
tot=0
for r in csv.reader(sys.stdin):
    print('   %-70s calls %s avg %.1f us' % (r[0].replace('void (anonymous namespace)::','')[:70], r[1], float(r[3])/1e3)); tot+=float(r[2])/1e3
print('   family total per 7 passes %.1f us' % tot)"
  rm -rf $out/$tag; }
one base
DET6D_GROUP_WAVES=8 one waves8
DET6D_GROUP_STREAM=0 one stream0
DET6D_GROUP_STREAM=3 one stream3
DET6D_LINEAR_PIPE=1 one linpipe
