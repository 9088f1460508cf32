# round 6: sequencers with the picks published after the decision loop: exactness on the sampler suites + us per pick (old = the
# previous commit's library)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
echo "== new"; python3 tests/gpu_scripts/fps_seq.py 2>&1 | grep -v amdgpu.ids | grep -v "m=[0-9]" | tail -12
python3 tests/gpu_scripts/fps_coop.py 2>&1 | grep -v amdgpu.ids | tail -9
echo "== old"; DET6D_KNOBS_LIB=$GRAFT_REPO_ROOT/scripts/r06/prev/libdet6d_hip_prev.so python3 tests/gpu_scripts/fps_coop.py quick 2>&1 | grep -v amdgpu.ids | tail -2
