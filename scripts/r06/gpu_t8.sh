# round 6: knobs of single launches replayed alone (knobs build): SA3's wide group (#8), the plain GEMMs (#6, 9, 11, 14)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24 DET6D_KNOBS_LIB=1
out=gpurun_out/r06_t8; mkdir -p $out
run() { tag=$1; shift; echo "== $tag"; env "$@" python3 scripts/r06/kernel_power.py uniform 0.4 $SEL 2>&1 | grep -v "amdgpu.ids\|^#"; }
SEL=7,8,12
run base X=1
run static DET6D_GROUP_STATIC=1
run nopre DET6D_GROUP_PRE=0
run sa3w8 DET6D_GROUP_SA3_WAVES=8
run s7 DET6D_GROUP_STREAM=7
run s0 DET6D_GROUP_STREAM=0
SEL=6,9,11,14
run base X=1
run t128x64 DET6D_LINEAR_K64MAX=4096 DET6D_LINEAR_N64MAX=1024
run nopipe DET6D_LINEAR_PIPE=0
run bk32 DET6D_LINEAR_BK32=64
run nbuf2 DET6D_LINEAR_NBUF2=1
