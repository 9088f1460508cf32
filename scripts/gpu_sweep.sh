cd $GRAFT_REPO_ROOT
for v in 0 1; do
  echo "== NBUF2=$v"
  if [ $v = 1 ]; then export DET6D_LINEAR_NBUF2=1; fi
  python scripts/gpu_linear_breakdown.py 2>&1 | tail -20 | cut -c1-62 | grep "K  512 N 1024\|K  256 N\|K  260 N  256\|K  128 N  256\|K  132 N  128\|total"
done
unset DET6D_LINEAR_NBUF2
DET6D_LINEAR_NBUF2=1 python -m pytest tests/test_ops_gpu.py -x -q -k "test_linear" 2>&1 | tail -1
