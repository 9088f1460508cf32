#!/bin/bash
mkdir -p gpurun_out/r04
for j in 1 4 64; do DET6D_EXPERIMENTS_LIB=1 DET6D_FPS_SEQ_PICKS=$j timeout 200 python scripts/experiments/gpu_fps_coop_sweep.py $1 2>&1 | grep -v amdgpu.ids; done 2>&1 | tee gpurun_out/r04/fps_coop_sweep$1.log
DET6D_EXPERIMENTS_LIB=1 DET6D_FPS_COOP_MULTI=0 timeout 200 python scripts/experiments/gpu_fps_coop_sweep.py $1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/fps_coop_sweep$1.log
