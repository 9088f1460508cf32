#!/bin/bash
# multi-pick sampler (experiments build): counters over the candidates per record and the cap on picks per round
mkdir -p gpurun_out/r04
for k in 2 4; do for j in 1 4 16; do
  DET6D_EXPERIMENTS_LIB=1 DET6D_FPS_SEQ=1 DET6D_FPS_SEQ_CANDS=$k DET6D_FPS_SEQ_PICKS=$j timeout 120 python scripts/experiments/gpu_fps_seq_stats.py $1 2>&1 | grep -v amdgpu.ids | sed "s/^/K=$k /"
done; done 2>&1 | tee gpurun_out/r04/fps_seq_stats$1.log
for k in 2 4; do
  echo "== K=$k"; DET6D_EXPERIMENTS_LIB=1 DET6D_FPS_SEQ=1 DET6D_FPS_SEQ_CANDS=$k timeout 300 python tests/gpu_scripts/fps_seq.py 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee gpurun_out/r04/fps_seq_k.log
