# round 6: score-weighted multi-pick sampler: the weighted-sampler tests, exactness + us per pick against the fat-thread kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "weight or fused_sampler" 2>&1 | tail -4
echo "== multi-pick"; python3 scripts/r06/sfps_time.py 2>&1 | grep -v amdgpu.ids
echo "== fat"; DET6D_KNOBS_LIB=1 DET6D_FPS_SEQW=0 python3 scripts/r06/sfps_time.py 2>&1 | grep -v amdgpu.ids
