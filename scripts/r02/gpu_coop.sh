cd $GRAFT_REPO_ROOT
timeout 900 python3 tests/gpu_scripts/fps_coop.py 2>&1 | grep -v amdgpu.ids
