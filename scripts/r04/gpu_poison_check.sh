#!/bin/bash
# the LDS-fill regression test: passes on the tree as it is, and (second half) FAILS when the round-4 fix is taken out again.
# The second half patches a COPY of the tree (round-4 advice: the first form of this script edited csrc/fps_coop.hip in place and
# never restored it, so the next build shipped the regression).
mkdir -p gpurun_out/r04
export DET6D_EXPERIMENTS_LIB=1 DET6D_DBG_POISON_LDS=0x7F7F0000
timeout 600 python tests/gpu_scripts/fps_coop.py 2>&1 | grep -v amdgpu.ids | tail -9
timeout 600 python tests/gpu_scripts/fps_seq.py 2>&1 | tail -11
DET6D_FPS_SEQ=1 timeout 600 python tests/gpu_scripts/fps_seq.py 2>&1 | tail -11
scratch=$(mktemp -d)
trap 'rm -rf "$scratch"' EXIT
cp -r de6d_amd include oracle tests "$scratch"/
sed -i 's/((unsigned)rec.k\[o\] & 0xFFFFu)/((unsigned)rec.k[o])/' "$scratch"/de6d_amd/csrc/fps_coop.hip
(cd "$scratch" && python -m de6d_amd._build --experiments 2>&1 | tail -2
 echo "--- without the mask (copy of the tree in $scratch):"
 timeout 600 python tests/gpu_scripts/fps_coop.py 2>&1 | grep -v amdgpu.ids | tail -9)
