#!/bin/bash
# multi-pick sampler: counters (experiments build) over the cap on picks per round
mkdir -p gpurun_out/r04
for j in 1 4 16; do
  DET6D_EXPERIMENTS_LIB=1 DET6D_FPS_SEQ=1 DET6D_FPS_SEQ_PICKS=$j timeout 120 python scripts/experiments/gpu_fps_seq_stats.py $1 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee gpurun_out/r04/fps_seq_stats$1.log
