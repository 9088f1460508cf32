#!/bin/bash
# look-ahead sampler: protocol counters (experiments build) over the poll-sleep / wake-up settings
mkdir -p gpurun_out/r04
for cfg in "4 1"; do
  set -- $cfg
  echo "== sleep $1 x 512 cycles, wakeup $2"
  DET6D_EXPERIMENTS_LIB=1 DET6D_FPS_SEQ=1 DET6D_FPS_SEQ_SLEEP=$1 DET6D_FPS_SEQ_WAKE=$2 timeout 120 python scripts/experiments/gpu_fps_seq_stats.py 2>&1 | grep -v amdgpu.ids
  echo
done 2>&1 | tee gpurun_out/r04/fps_seq_stats.log
