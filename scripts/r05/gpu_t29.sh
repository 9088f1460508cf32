# round 5: grid caps of the row-stack and register-chain kernels at the 80-scene pass size (experiments build: DET6D_ROWS_BLOCKS,
# DET6D_CHAIN_BLOCKS); eager per-kernel averages
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24 DET6D_EXPERIMENTS_LIB=1
out=gpurun_out/r05_t29; mkdir -p $out
A="--steps 5 --warmup 2 --batch 80 --streams 1 --no-graph --cpu-scenes 0 --no-roofline --no-legs --preroll 0 --windows 1"
one() { tag=$1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -o k -- python3 bench.py $A > $out/$tag.log 2>&1
  f=$(find $out/$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag"; grep "mlp_rows_kernel\|mlp_chain_reg_kernel" $f | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin): print('   %-60s calls %s avg %.1f us' % (r[0].replace('void (anonymous namespace)::','')[:60], r[1], float(r[3])/1e3))"
  rm -rf $out/$tag; }
one base
for rb in 512 768 1536 2048; do export DET6D_ROWS_BLOCKS=$rb; one rows_$rb; done; unset DET6D_ROWS_BLOCKS
for cb in 768 1024 1536 3072 4096; do export DET6D_CHAIN_BLOCKS=$cb; one chain_$cb; done
