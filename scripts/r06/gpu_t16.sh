# round 6: staggered start of the workgroups that share a CU in the group kernels (knobs build): launches replayed alone
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24 DET6D_KNOBS_LIB=1
run() { tag=$1; shift; echo "== $tag"; env "$@" python3 scripts/r06/kernel_power.py uniform 0.3 7,8,12,13 2>&1 | grep -v "amdgpu.ids\|^#"; }
for d in 0 4 8 12 16 24; do run stagger_$d DET6D_GROUP_STAGGER=$d; done
