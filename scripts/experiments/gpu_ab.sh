# A/B of the GEMM family: current libdet6d_hip.so vs libdet6d_hip_old.so on the same box
cd $GRAFT_REPO_ROOT
python scripts/gpu_linear_breakdown.py 2>&1 | tail -36 > gpurun_out/bd_new.txt
cp de6d_amd/csrc/libdet6d_hip.so /tmp/new.so
cp de6d_amd/csrc/libdet6d_hip_old.so de6d_amd/csrc/libdet6d_hip.so
python scripts/gpu_linear_breakdown.py 2>&1 | tail -36 > gpurun_out/bd_old.txt
cp /tmp/new.so de6d_amd/csrc/libdet6d_hip.so
paste <(cut -c1-62 gpurun_out/bd_new.txt) <(cut -c30-62 gpurun_out/bd_old.txt)
tail -1 gpurun_out/bd_old.txt
