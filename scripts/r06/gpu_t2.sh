# round 6: det6d_linear variants on the four plain GEMM shapes of an 80-scene pass (experiments build; idle chip, 20 launches each)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export DET6D_EXPERIMENTS_LIB=1
out=gpurun_out/r06_t2; mkdir -p $out
run() { echo "== $1"; env TAG="$1" $2 python3 scripts/r06/linear_shapes.py 2>&1 | tail -5; }
run base ""
run pipe "DET6D_LINEAR_PIPE=1"
run pipe_nbuf2 "DET6D_LINEAR_PIPE=1 DET6D_LINEAR_NBUF2=1"
run nbuf2 "DET6D_LINEAR_NBUF2=1"
run pipe_bk32 "DET6D_LINEAR_PIPE=1 DET6D_LINEAR_BK32=64"
run t128x64 "DET6D_LINEAR_K64MAX=4096 DET6D_LINEAR_N64MAX=1024"
run pipe_t128x64 "DET6D_LINEAR_PIPE=1 DET6D_LINEAR_K64MAX=4096 DET6D_LINEAR_N64MAX=1024"
run base_again ""
