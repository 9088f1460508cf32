#!/bin/bash
# 16384-point D-FPS: exactness + us per pick, shipped wave-skip sampler and the experiments build's multi-pick sampler
mkdir -p gpurun_out/r04
timeout 300 python tests/gpu_scripts/fps_seq.py $1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/fps_skip.log
DET6D_EXPERIMENTS_LIB=1 DET6D_FPS_SEQ=1 timeout 300 python tests/gpu_scripts/fps_seq.py $1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/fps_seq.log
