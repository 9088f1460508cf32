# round 6: is SA3's wide group (#8) / the head's narrow group bound by the L2 -> CU weight stream?  experiments build,
# DET6D_GROUP_WHATIF=1 serves every B fragment from the first two weight rows (wrong results, timing only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24 DET6D_EXPERIMENTS_LIB=1 DET6D_GROUP_PHASE_C3=7
run() { tag=$1; shift; echo "== $tag"; env "$@" python3 scripts/r06/kernel_power.py uniform 0.4 7,8,12,13 2>&1 | grep -v "amdgpu.ids\|^#"; }
run base DET6D_GROUP_STREAM=0
run whatif DET6D_GROUP_STREAM=0 DET6D_GROUP_WHATIF=1
