# round 6: clock + power of every GEMM-family launch replayed alone (review item 2)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r06_t7; mkdir -p $out
python3 scripts/r06/kernel_power.py uniform 1.0 2>&1 | tee $out/kernel_power_uniform.txt | tail -20
DET6D_KNOBS_LIB=1 DET6D_GROUP_STREAM=2 python3 scripts/r06/kernel_power.py uniform 1.0 2>&1 | tee $out/kernel_power_uniform_onepass.txt | tail -20
