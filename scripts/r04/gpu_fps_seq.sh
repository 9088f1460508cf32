#!/bin/bash
# look-ahead sampler: exactness + us/round for every region granularity, beside the wave-skip sampler
mkdir -p gpurun_out/r04
for s in 1 0; do
  DET6D_FPS_SEQ=$s timeout 300 python tests/gpu_scripts/fps_seq.py $1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/fps_seq_$s.log
done
