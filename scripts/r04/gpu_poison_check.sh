#!/bin/bash
# the LDS-fill regression test: passes on the tree as it is, and (second half) FAILS when the round-4 fix is taken out again
mkdir -p gpurun_out/r04
export DET6D_EXPERIMENTS_LIB=1 DET6D_DBG_POISON_LDS=0x7F7F0000
timeout 600 python tests/gpu_scripts/fps_coop.py 2>&1 | grep -v amdgpu.ids | tail -9
timeout 600 python tests/gpu_scripts/fps_seq.py 2>&1 | tail -11
DET6D_FPS_SEQ=1 timeout 600 python tests/gpu_scripts/fps_seq.py 2>&1 | tail -11
sed -i 's/((unsigned)rec.k\[o\] & 0xFFFFu)/((unsigned)rec.k[o])/' de6d_amd/csrc/fps_coop.hip
python -m de6d_amd._build --experiments 2>&1 | tail -2
echo "--- without the mask:"
timeout 600 python tests/gpu_scripts/fps_coop.py 2>&1 | grep -v amdgpu.ids | tail -9
