# round 6: phase timers of the group kernels (experiments build), launches replayed back to back
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
DET6D_GROUP_PHASE_C2=256 DET6D_GROUP_PHASE_C3=256 python3 scripts/r06/group_phase.py 8 2>&1 | grep -v amdgpu.ids
DET6D_GROUP_PHASE_C2=128 DET6D_GROUP_PHASE_C3=256 python3 scripts/r06/group_phase.py 7 2>&1 | grep -v amdgpu.ids
python3 scripts/r06/group_phase.py 13 2>&1 | grep -v amdgpu.ids
DET6D_GROUP_PHASE_C2=256 DET6D_GROUP_PHASE_C3=512 python3 scripts/r06/group_phase.py 12 2>&1 | grep -v amdgpu.ids
