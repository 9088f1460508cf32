# round 6: sampler decision loop with one exit: exactness + us per pick
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
python3 tests/gpu_scripts/fps_seq.py > gpurun_out/r06_t22_fps_seq.txt 2>&1; grep "b=8 n=16384 m=4096\|b=8 n=4096 m=512\|b=32\|ALL\|False" gpurun_out/r06_t22_fps_seq.txt
python3 tests/gpu_scripts/fps_coop.py 2>&1 | grep -v amdgpu.ids | tail -9
