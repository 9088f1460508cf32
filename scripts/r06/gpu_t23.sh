# round 6: counters of the first-layer sampler after this round's sequencer changes (experiments build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export DET6D_EXPERIMENTS_LIB=1
python3 scripts/experiments/gpu_fps_seq_stats.py 2>&1 | grep -v amdgpu.ids
python3 scripts/experiments/gpu_fps_seq_stats.py beam 2>&1 | grep -v amdgpu.ids
