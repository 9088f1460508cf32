#!/bin/bash
mkdir -p gpurun_out/r04
for s in 1; do
  DET6D_EXPERIMENTS_LIB=1 DET6D_FPS_SEQ=$s timeout 120 python scripts/experiments/gpu_fps_seq_stats.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/fps_seq_stats_$s.log
done
